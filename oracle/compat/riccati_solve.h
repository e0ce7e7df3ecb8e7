/* riccati_solve.h -- source-compatibility forward for the reference's test programs (test infrastructure:
 * see riccati_compat.h). */
#include "riccati_compat.h"
