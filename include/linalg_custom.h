/* linalg_custom.h -- source-compatibility forward: the clap_* names of the reference's internal
 * backend (src/linalg_custom.h) are declared in ndlqr.h. */
#include "ndlqr.h"
