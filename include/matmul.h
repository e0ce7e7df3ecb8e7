/* matmul.h -- source-compatibility forward. The reference's src/matmul.h declares experimental AVX2
 * micro-kernels that are in no build target (src/CMakeLists.txt:85-86); callers that still include the
 * header (test/parallel_test.c:5) only use MatrixMultiply, which lives in ndlqr.h. */
#include "ndlqr.h"
