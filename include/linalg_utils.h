/* linalg_utils.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/linalg_utils.h and friends). */
#include "ndlqr.h"
