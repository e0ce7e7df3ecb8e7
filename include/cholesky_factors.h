/* cholesky_factors.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/cholesky_factors.h and friends). */
#include "ndlqr.h"
