/* lqr_data.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/lqr_data.h and friends). */
#include "ndlqr.h"
