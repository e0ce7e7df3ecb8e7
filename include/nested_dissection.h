/* nested_dissection.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/nested_dissection.h and friends). */
#include "ndlqr.h"
