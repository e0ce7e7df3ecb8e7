/* Pulls in the reference's own public headers (found via -I/root/reference/src at build
 * time; nothing is copied). The umbrella ndlqr.h is avoided because it drags in
 * matmul.h -> eigen_c/eigen_c.h, which are dead / not built (src/CMakeLists.txt:85-86).
 * solver.h already includes binary_tree.h, cholesky_factors.h (which has no include
 * guard, so it must not be included twice), linalg.h, lqr_problem.h and nddata.h. */
#pragma once
#include "nested_dissection.h"
#include "solve.h"
#include "solver.h"
