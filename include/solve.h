/* solve.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/solve.h and friends). */
#include "ndlqr.h"
