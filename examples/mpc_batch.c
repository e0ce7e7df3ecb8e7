/*
 * mpc_batch.c -- the batch API in an MPC-style loop: many independent LQR problems factored once
 * on the GPU (NDLQR_FLAG_KEEP_RECORDS / NDLQR_FLAG_KEEP_FACT), then re-solved for new right-hand sides (new initial state
 * and linear cost / dynamics offsets, same A, B, Q, R) without refactoring. Every solution is
 * checked on the device against its raw problem data (KKT residual).
 *
 *   gcc -Iinclude examples/mpc_batch.c -Lrslqr_amd -lrslqr_amd -Wl,-rpath,$PWD/rslqr_amd -lm -o mpc_batch
 *   ./mpc_batch [nstates ninputs nhorizon batch steps]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ndlqr.h"

static double worst_relative_residual(NdLqrBatchSolver* bs, int batch, double* res, double* bn) {
  double worst = 0.0;
  if (ndlqr_BatchKktResiduals(bs, res, bn) != 0) return -1.0;
  for (int p = 0; p < batch; ++p) {
    const double rel = res[p] / (bn[p] > 1.0 ? bn[p] : 1.0);
    if (rel > worst) worst = rel;
  }
  return worst;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12, m = argc > 2 ? atoi(argv[2]) : 4;
  const int N = argc > 3 ? atoi(argv[3]) : 64, batch = argc > 4 ? atoi(argv[4]) : 32;
  const int steps = argc > 5 ? atoi(argv[5]) : 5;

  NdLqrBatchSolver* bs = ndlqr_NewBatchSolver(n, m, N, batch, -1);
  if (!bs) { fprintf(stderr, "no solver (is a HIP device visible?)\n"); return 2; }
  /* keep what a re-solve needs: the separator records are enough on size-specialised shapes
   * (cheaper than the whole factor array); other shapes fall back to NDLQR_FLAG_KEEP_FACT below */
  ndlqr_BatchSetFlags(bs, NDLQR_FLAG_KEEP_RECORDS);

  /* flat host arrays of the whole batch: A [batch][N][n*n] (column-major per knot), B, Q, R, q, r, d, x0 */
  const size_t sA = (size_t)N * n * n, sB = (size_t)N * n * m, sn = (size_t)N * n, sm = (size_t)N * m;
  double* A = malloc(sizeof(double) * batch * sA); double* B = malloc(sizeof(double) * batch * sB);
  double* Q = malloc(sizeof(double) * batch * sn); double* R = malloc(sizeof(double) * batch * sm);
  double* q = malloc(sizeof(double) * batch * sn); double* r = malloc(sizeof(double) * batch * sm);
  double* d = malloc(sizeof(double) * batch * sn); double* x0 = malloc(sizeof(double) * batch * n);
  double* res = malloc(sizeof(double) * batch); double* bn = malloc(sizeof(double) * batch);
  for (int p = 0; p < batch; ++p)
    if (ndlqr_GenerateSyntheticFlat(n, m, N, 1 + (uint64_t)p, A + p * sA, B + p * sB, Q + p * sn, R + p * sm,
                                    q + p * sn, r + p * sm, d + p * sn, x0 + p * n) != 0) return 3;

  if (ndlqr_InitializeBatchFlat(bs, A, B, Q, R, q, r, d, x0) != 0) return 4;
  if (ndlqr_SolveBatch(bs) != 0) return 5;  /* factor + solve, factorisation stays on the device */
  printf("factor + solve : %8.3f ms for %d problems of (n=%d, m=%d, N=%d), worst KKT residual %.2e\n",
         ndlqr_BatchSolveTimeMs(bs), batch, n, m, N, worst_relative_residual(bs, batch, res, bn));

  const int nvars = ndlqr_BatchNumVars(bs);
  double* soln = malloc(sizeof(double) * (size_t)batch * nvars);
  for (int it = 0; it < steps; ++it) {
    /* "receding horizon": start every problem from the state its last plan reached after one
     * step (x_1 of knot 1 sits behind lambda_1, x_0, u_0, lambda_2 in the solution vector) */
    if (ndlqr_CopyBatchSolutions(bs, soln) != nvars) return 6;
    for (int p = 0; p < batch; ++p)
      memcpy(x0 + (size_t)p * n, soln + (size_t)p * nvars + (2 * n + m) + n, sizeof(double) * n);
    if (ndlqr_BatchSetRhsFlat(bs, q, r, d, x0) != 0) return 7;
    if (ndlqr_SolveBatchRhsOnly(bs) != 0) {
      if (it > 0) return 8;
      /* no record-based re-solve for this shape: keep the factor array instead and factor again */
      ndlqr_BatchSetFlags(bs, NDLQR_FLAG_KEEP_FACT);
      if (ndlqr_SolveBatch(bs) != 0) return 8;
      if (ndlqr_SolveBatchRhsOnly(bs) != 0) return 8;
    }
    const double worst = worst_relative_residual(bs, batch, res, bn);
    printf("re-solve %2d    : %8.3f ms, worst KKT residual %.2e\n", it, ndlqr_BatchSolveTimeMs(bs), worst);
    if (!(worst < 1e-9)) return 9;
  }
  ndlqr_FreeBatchSolver(bs);
  free(A); free(B); free(Q); free(R); free(q); free(r); free(d); free(x0); free(res); free(bn); free(soln);
  return 0;
}
