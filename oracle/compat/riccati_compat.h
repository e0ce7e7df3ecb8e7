/* riccati_compat.h -- TEST INFRASTRUCTURE, not part of the product library.
 *
 * The reference's serial Riccati comparison solver (src/riccati_solver.h:44-178,
 * src/riccati_solve.h:25-49) is outside the hot path this repository implements (SURVEY.md section 2,
 * row 11). Two of the reference's own test programs (test/riccati_solver_test.c,
 * test/sample_problem_test.c) call it next to ndlqr_Solve, so a source-compatible stand-in lives here,
 * composed from the product's device-backed Matrix* helpers (oracle/compat/riccati_compat.c), and is
 * linked into those two test binaries only (oracle/Makefile). librslqr_amd.so neither contains nor
 * needs it. Field order and the layout of `data` are the reference's (its tests read them directly).
 */
#ifndef NDLQR_RICCATI_COMPAT_H_
#define NDLQR_RICCATI_COMPAT_H_
#include "ndlqr.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  LQRProblem* prob;
  int nhorizon, nstates, ninputs, nvars;
  double* data;  /* P_k p_k K_k d_k per knot | solution [y0 x0 u0 y1 ...] | temporaries x2 */
  Matrix* K;     /* N-1 feedback gains (m x n) */
  Matrix* d;     /* N-1 feedforward terms (m) */
  Matrix* P;     /* N cost-to-go Hessians (n x n) */
  Matrix* p;     /* N cost-to-go gradients (n) */
  Matrix* X;     /* N states */
  Matrix* U;     /* N-1 inputs */
  Matrix* Y;     /* N multipliers */
  Matrix* Qx;    /* action-value temporaries, two of each */
  Matrix* Qu;
  Matrix* Qxx;
  Matrix* Qux;
  Matrix* Quu;
  double t_solve_ms, t_backward_pass_ms, t_forward_pass_ms;
} RiccatiSolver;

RiccatiSolver* ndlqr_NewRiccatiSolver(LQRProblem* lqrprob);
int ndlqr_FreeRiccatiSolver(RiccatiSolver* solver);
int ndlqr_PrintRiccatiSummary(RiccatiSolver* solver);
Matrix ndlqr_GetRiccatiSolution(RiccatiSolver* solver);
int ndlqr_GetNumVarsRiccati(RiccatiSolver* solver);
int ndlqr_CopyRiccatiSolution(RiccatiSolver* solver, double* soln);
int ndlqr_GetRiccatiSolveTimes(RiccatiSolver* solver, double* t_solve, double* t_bp, double* t_fp);
int ndlqr_SolveRiccati(RiccatiSolver* solver);
int ndlqr_BackwardPass(RiccatiSolver* solver);
int ndlqr_ForwardPass(RiccatiSolver* solver);

#ifdef __cplusplus
}
#endif
#endif /* NDLQR_RICCATI_COMPAT_H_ */
