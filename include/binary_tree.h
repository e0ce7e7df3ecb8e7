/* binary_tree.h -- source-compatibility forward: the whole ndlqr API lives in ndlqr.h (the
 * reference splits it across src/binary_tree.h and friends). */
#include "ndlqr.h"
