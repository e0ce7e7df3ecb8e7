/* lqr_problem.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/lqr_problem.h and friends). */
#include "ndlqr.h"
