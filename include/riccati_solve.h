/* riccati_solve.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/riccati_solve.h and friends). */
#include "ndlqr.h"
