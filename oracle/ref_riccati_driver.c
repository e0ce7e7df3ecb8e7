/*
 * ref_riccati_driver.c -- driver for the reference's serial Riccati comparison solver
 * (/root/reference/src/riccati_solver.c, riccati_solve.c), linked into oracle/_ref/libref.so.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/Makefile). SURVEY.md 8(f)-4: Riccati is the reference's own
 * second CPU column (docs/Overview.dox:33-38, test/sample_problem_test.c:127-177); bench.py times it
 * next to the reference's ndlqr_Solve as `cpu_baseline.riccati`, ON THE TWO JSON FIXTURES ONLY -- it
 * diverges on random long horizons, so it is never used as a checker off-fixture.
 *
 * The two Riccati files compile from where they lie with gcc alone. They read the problem through
 * the seven one-line views ndlqr_GetA ... ndlqr_Getr of the problem container
 * (/root/reference/src/lqr_data.c:74-107). That file is not compilable here (it includes cJSON's
 * header, absent from this image, for the JSON half of the container) and -- like the rest of the
 * problem containers -- is not part of any solve path; this driver, which already fills the plain
 * LQRData / LQRProblem structs itself (make_problem in ref_driver.c), supplies those seven views of
 * its own structs. Everything that is timed (ndlqr_BackwardPass, ndlqr_ForwardPass, the dense
 * helpers under them) is the reference's code.
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime under -std=c11 (the reference's dialect: it keeps fp contraction off) */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "riccati_solve.h"   /* the reference's own header (-I/root/reference/src) */

Matrix ndlqr_GetA(LQRData* l) { Matrix v = {l->nstates, l->nstates, l->A}; return v; }
Matrix ndlqr_GetB(LQRData* l) { Matrix v = {l->nstates, l->ninputs, l->B}; return v; }
Matrix ndlqr_Getd(LQRData* l) { Matrix v = {l->nstates, 1, l->d}; return v; }
Matrix ndlqr_GetQ(LQRData* l) { Matrix v = {l->nstates, 1, l->Q}; return v; }
Matrix ndlqr_Getq(LQRData* l) { Matrix v = {l->nstates, 1, l->q}; return v; }
Matrix ndlqr_GetR(LQRData* l) { Matrix v = {l->ninputs, 1, l->R}; return v; }
Matrix ndlqr_Getr(LQRData* l) { Matrix v = {l->ninputs, 1, l->r}; return v; }

LQRProblem* ref_make_problem(int n, int m, int N, const double* A, const double* B, const double* Q,
                             const double* R, const double* q, const double* r, const double* d,
                             const double* x0);
void ref_free_problem(LQRProblem* p);

/* ndlqr_SolveRiccati `reps` times on one problem; soln (nvars doubles, may be NULL) receives the
 * solution vector in the reference's [lambda x u] order (src/riccati_solver.c:167-177). Returns the
 * mean wall ms of a solve (CLOCK_MONOTONIC around the call; the solver's own clock() timer measures
 * the same on one thread) or -1. */
double ref_riccati_bench(int n, int m, int N, int reps, const double* A, const double* B,
                         const double* Q, const double* R, const double* q, const double* r,
                         const double* d, const double* x0, double* soln, int* nvars_out) {
  LQRProblem* p = ref_make_problem(n, m, N, A, B, Q, R, q, r, d, x0);
  RiccatiSolver* s = ndlqr_NewRiccatiSolver(p);
  if (!s) { ref_free_problem(p); return -1.0; }
  if (ndlqr_SolveRiccati(s) != 0) { ndlqr_FreeRiccatiSolver(s); ref_free_problem(p); return -1.0; }  /* warm-up */
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < reps; ++i) ndlqr_SolveRiccati(s);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const int nvars = ndlqr_GetNumVarsRiccati(s);
  if (nvars_out) *nvars_out = nvars;
  if (soln) ndlqr_CopyRiccatiSolution(s, soln);
  ndlqr_FreeRiccatiSolver(s);
  ref_free_problem(p);
  return ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / (reps > 0 ? reps : 1);
}
