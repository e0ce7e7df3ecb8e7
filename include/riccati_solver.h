/* riccati_solver.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/riccati_solver.h and friends). */
#include "ndlqr.h"
