/*
 * mpc_step.c -- the asynchronous MPC step of the batch API in plain C: per iteration the new initial states go up, the
 * whole batch is factored and solved, and ONLY what the loop consumes comes down -- u of knot 0 of every problem
 * (ndlqr_BatchSetStepSelection) -- with two steps in flight on the two buffer sets of the solver. What the reference
 * does per iteration with ndlqr_ResetSolver + ndlqr_InitializeWithLQRProblem + ndlqr_Solve + ndlqr_CopySolution
 * (src/solve.h:20-32), for `batch` independent problems at once.
 *
 *   gcc -Iinclude examples/mpc_step.c -Lrslqr_amd -lrslqr_amd -Wl,-rpath,$PWD/rslqr_amd -lm -o mpc_step
 *   ./mpc_step [nstates ninputs nhorizon batch steps [keep [only]]]
 *
 * keep = 1: NDLQR_FLAG_KEEP_RECORDS -- a step never changes A, B, Q, R, so only the first one factors and every further
 * step is the right-hand-side re-solve on the kept records (0.48 instead of 0.63 ms per step of 1024 x (12,4,256)).
 * only = 1: NDLQR_SOLN_ONLY -- the steps COMPUTE nothing but u of knot 0 (the last launch of the back-substitution runs
 * one workgroup per problem instead of N / 8): 0.43 ms per step, 0.30 with keep = 1.
 *
 * Each step starts every problem from x_1 = A_0 x_0 + B_0 u_0 + d_0 of the step before the previous one (the loop runs
 * one step behind the solver: the freshest u_0 it may read is that of step it - 1 while step it is in flight).
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ndlqr.h"

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12, m = argc > 2 ? atoi(argv[2]) : 4;
  const int N = argc > 3 ? atoi(argv[3]) : 64, batch = argc > 4 ? atoi(argv[4]) : 256;
  const int steps = argc > 5 ? atoi(argv[5]) : 20;
  const int keep = argc > 6 ? atoi(argv[6]) : 0;
  const int only = argc > 7 ? atoi(argv[7]) : 0;

  NdLqrBatchSolver* bs = ndlqr_NewBatchSolver(n, m, N, batch, -1);
  if (!bs) { fprintf(stderr, "no solver (is a HIP device visible?)\n"); return 2; }
  if (keep && ndlqr_BatchSetFlags(bs, NDLQR_FLAG_KEEP_RECORDS) != 0) return 2;
  const size_t sA = (size_t)N * n * n, sB = (size_t)N * n * m, sn = (size_t)N * n, sm = (size_t)N * m;
  double* A = malloc(sizeof(double) * batch * sA); double* B = malloc(sizeof(double) * batch * sB);
  double* Q = malloc(sizeof(double) * batch * sn); double* R = malloc(sizeof(double) * batch * sm);
  double* q = malloc(sizeof(double) * batch * sn); double* r = malloc(sizeof(double) * batch * sm);
  double* d = malloc(sizeof(double) * batch * sn); double* x0 = malloc(sizeof(double) * batch * n);
  for (int p = 0; p < batch; ++p)
    if (ndlqr_GenerateSyntheticFlat(n, m, N, 1 + (uint64_t)p, A + p * sA, B + p * sB, Q + p * sn, R + p * sm,
                                    q + p * sn, r + p * sm, d + p * sn, x0 + p * n) != 0) return 3;
  if (ndlqr_InitializeBatchFlat(bs, A, B, Q, R, q, r, d, x0) != 0) return 4;  /* A, B, Q, R, q, r, d stay resident */

  /* pinned host memory keeps the copies asynchronous: x0 going up, u_0 coming down, one pair per step in flight */
  double* xs[2] = {ndlqr_HostAlloc(sizeof(double) * batch * n), ndlqr_HostAlloc(sizeof(double) * batch * n)};
  double* u0[2] = {ndlqr_HostAlloc(sizeof(double) * batch * m), ndlqr_HostAlloc(sizeof(double) * batch * m)};
  if (!xs[0] || !xs[1] || !u0[0] || !u0[1]) return 5;
  memcpy(xs[0], x0, sizeof(double) * batch * n);
  memcpy(xs[1], x0, sizeof(double) * batch * n);
  /* bring down u of knot 0 alone (only: and compute nothing else) */
  if (ndlqr_BatchSetStepSelection(bs, 0, 1, NDLQR_SOLN_INPUT | (only ? NDLQR_SOLN_ONLY : 0u)) != 0) return 6;

  double* res = malloc(sizeof(double) * batch); double* bn = malloc(sizeof(double) * batch);
  const double t0 = now_ms();
  for (int it = 0; it < steps; ++it) {
    double* x = xs[it & 1];
    if (it >= 2) {
      /* step it - 2 is complete (waited for below in iteration it - 1): advance its problems by one knot */
      const double* up = u0[it & 1];  /* written by step it - 2 */
      for (int p = 0; p < batch; ++p) {
        const double *Ap = A + p * sA, *Bp = B + p * sB, *dp = d + p * sn, *xo = xs[it & 1] + (size_t)p * n;
        double xn[64];
        for (int i = 0; i < n && i < 64; ++i) {
          double acc = dp[i];
          for (int j = 0; j < n; ++j) acc += Ap[i + n * j] * xo[j];
          for (int j = 0; j < m; ++j) acc += Bp[i + n * j] * up[(size_t)p * m + j];
          xn[i] = acc;
        }
        memcpy(x + (size_t)p * n, xn, sizeof(double) * (n < 64 ? n : 64));
      }
    }
    if (ndlqr_BatchStepAsync(bs, NULL, NULL, NULL, x, u0[it & 1]) != 0) return 7;  /* q, r, d: unchanged */
    if (it >= 1) {
      const int err = ndlqr_BatchSynchronizePrevious(bs);  /* u0[(it - 1) & 1] is complete now */
      if (err != 0) { fprintf(stderr, "step %d: %d\n", it - 1, err); return 8; }
    }
  }
  if (ndlqr_BatchSynchronize(bs) != 0) return 9;
  const double ms = (now_ms() - t0) / steps;
  double* u_last = malloc(sizeof(double) * batch * m);
  memcpy(u_last, u0[(steps - 1) & 1], sizeof(double) * batch * m);
  if (only) {
    /* the solver holds u_0 alone (the whole vector is refused: -1); the last step once more without the bit brings
     * everything back for the checks below */
    if (ndlqr_BatchKktResiduals(bs, res, bn) == 0) return 13;
    if (ndlqr_BatchSetStepSelection(bs, 0, 1, NDLQR_SOLN_INPUT) != 0) return 6;
    if (ndlqr_BatchStepAsync(bs, NULL, NULL, NULL, xs[(steps - 1) & 1], u0[(steps - 1) & 1]) != 0) return 7;
    if (ndlqr_BatchSynchronize(bs) != 0) return 9;
  }
  /* the solution resident on the device is the last step's: check every problem against its raw data */
  if (ndlqr_BatchKktResiduals(bs, res, bn) != 0) return 10;
  double worst = 0.0;
  for (int p = 0; p < batch; ++p) {
    const double rel = res[p] / (bn[p] > 1.0 ? bn[p] : 1.0);
    if (rel > worst) worst = rel;
  }
  /* ... and the slice a step brought down is that slice of the resident solution */
  double* u_chk = malloc(sizeof(double) * batch * m);
  if (ndlqr_CopyBatchSolutionSlices(bs, 0, 1, NDLQR_SOLN_INPUT, u_chk) != 0) return 11;
  const int same = memcmp(u_chk, u_last, sizeof(double) * batch * m) == 0;
  printf("%d MPC steps of %d problems (n=%d, m=%d, N=%d): %.3f ms per step end to end, worst KKT residual %.2e, "
         "u_0 of the last step %s\n", steps, batch, n, m, N, ms, worst, same ? "matches the resident solution" : "DIFFERS");
  ndlqr_HostFree(xs[0]); ndlqr_HostFree(xs[1]); ndlqr_HostFree(u0[0]); ndlqr_HostFree(u0[1]);
  ndlqr_FreeBatchSolver(bs);
  free(A); free(B); free(Q); free(R); free(q); free(r); free(d); free(x0); free(res); free(bn); free(u_chk); free(u_last);
  return (worst < 1e-9 && same) ? 0 : 12;
}
