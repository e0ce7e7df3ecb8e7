/* solver.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/solver.h and friends). */
#include "ndlqr.h"
