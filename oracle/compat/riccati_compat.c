/*
 * riccati_compat.c -- TEST INFRASTRUCTURE (see riccati_compat.h): stand-in for the reference's serial
 * Riccati comparison solver (src/riccati_solver.h:44-178, src/riccati_solve.h:25-49), linked only into
 * the reference test programs that call it. Not part of librslqr_amd.so.
 *
 * Not part of the hot path and not a fallback for it: a second, independent solver for the same
 * LQR problem that the reference ships as a comparison (test/sample_problem_test.c prints rsLQR
 * against it, test/riccati_solver_test.c pins its intermediate quantities). The recursion is
 * composed from this library's Matrix helpers, i.e. every product and Cholesky solve runs through
 * the device shim like the rest of the dense helper API -- small launches with host round trips,
 * meant for the reference's tests and examples, not for throughput.
 *
 * Caller-visible layout (the reference's tests rely on it, riccati_solver_test.c:23-26,319-343):
 * one block of doubles holding, in this order, per knot P_k, p_k (and K_k, d_k for k < N-1),
 * then the solution vector [y_0 x_0 u_0 y_1 x_1 u_1 ... y_{N-1} x_{N-1}] -- the same ordering as
 * ndlqr_GetSolution --, then two sets of the action-value temporaries Qx, Qu, Qxx, Qux, Quu.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "riccati_compat.h"

static Matrix carve(double* base, size_t* cursor, int rows, int cols) {
  Matrix m = {rows, cols, base + *cursor};
  *cursor += (size_t)rows * cols;
  return m;
}

RiccatiSolver* ndlqr_NewRiccatiSolver(LQRProblem* lqrprob) {
  if (!lqrprob || lqrprob->nhorizon < 1) return NULL;
  const int N = lqrprob->nhorizon;
  const int n = lqrprob->lqrdata[0]->nstates, m = lqrprob->lqrdata[0]->ninputs;
  const size_t gains = (size_t)(N - 1) * ((size_t)m * n + m);
  const size_t costs = (size_t)N * ((size_t)n * n + n);
  const size_t soln = (size_t)N * 2 * n + (size_t)(N - 1) * m;
  const size_t qset = (size_t)n + m + (size_t)n * n + (size_t)m * n + (size_t)m * m;
  const size_t total = gains + costs + soln + 2 * qset;

  RiccatiSolver* s = (RiccatiSolver*)calloc(1, sizeof(RiccatiSolver));
  double* data = (double*)calloc(total, sizeof(double));
  Matrix* knot = (Matrix*)malloc(sizeof(Matrix) * (size_t)(7 * N));  /* P p X Y (N each), K d U (N-1 each) */
  Matrix* q = (Matrix*)malloc(sizeof(Matrix) * 10);
  if (!s || !data || !knot || !q) { free(s); free(data); free(knot); free(q); return NULL; }
  s->P = knot; s->p = knot + N; s->X = knot + 2 * N; s->Y = knot + 3 * N;
  s->K = knot + 4 * N; s->d = knot + 5 * N; s->U = knot + 6 * N;

  size_t at = 0;
  for (int k = 0; k < N; ++k) {
    s->P[k] = carve(data, &at, n, n);
    s->p[k] = carve(data, &at, n, 1);
    if (k < N - 1) {
      s->K[k] = carve(data, &at, m, n);
      s->d[k] = carve(data, &at, m, 1);
    }
  }
  for (int k = 0; k < N; ++k) {  /* the solution vector, contiguous */
    s->Y[k] = carve(data, &at, n, 1);
    s->X[k] = carve(data, &at, n, 1);
    if (k < N - 1) s->U[k] = carve(data, &at, m, 1);
  }
  s->Qx = q; s->Qu = q + 2; s->Qxx = q + 4; s->Qux = q + 6; s->Quu = q + 8;
  for (int i = 0; i < 2; ++i) {
    s->Qx[i] = carve(data, &at, n, 1);
    s->Qu[i] = carve(data, &at, m, 1);
    s->Qxx[i] = carve(data, &at, n, n);
    s->Qux[i] = carve(data, &at, m, n);
    s->Quu[i] = carve(data, &at, m, m);
  }
  s->prob = lqrprob;
  s->nhorizon = N; s->nstates = n; s->ninputs = m;
  s->nvars = (2 * n + m) * N - m;
  s->data = data;
  return s;
}

int ndlqr_FreeRiccatiSolver(RiccatiSolver* solver) {
  if (!solver) return -1;
  free(solver->data);
  free(solver->P);   /* one allocation for all per-knot views */
  free(solver->Qx);  /* one allocation for the temporaries' views */
  free(solver);
  return 0;
}

int ndlqr_PrintRiccatiSummary(RiccatiSolver* solver) {
  if (!solver) return -1;
  const double t = solver->t_solve_ms;
  printf("Riccati solve summary (dense helpers on the device)\n");
  printf("  total:         %.2f ms\n", t);
  printf("  backward pass: %.2f ms (%.1f %%)\n", solver->t_backward_pass_ms,
         t > 0 ? 100.0 * solver->t_backward_pass_ms / t : 0.0);
  printf("  forward pass:  %.2f ms (%.1f %%)\n", solver->t_forward_pass_ms,
         t > 0 ? 100.0 * solver->t_forward_pass_ms / t : 0.0);
  return 0;
}

Matrix ndlqr_GetRiccatiSolution(RiccatiSolver* solver) {
  Matrix none = {0, 0, NULL};
  if (!solver) return none;
  Matrix soln = {solver->nvars, 1, solver->Y[0].data};
  return soln;
}

int ndlqr_GetNumVarsRiccati(RiccatiSolver* solver) { return solver ? solver->nvars : -1; }

int ndlqr_CopyRiccatiSolution(RiccatiSolver* solver, double* soln) {
  if (!solver || !soln) return -1;
  memcpy(soln, solver->Y[0].data, sizeof(double) * (size_t)solver->nvars);
  return solver->nvars;
}

int ndlqr_GetRiccatiSolveTimes(RiccatiSolver* solver, double* t_solve, double* t_bp, double* t_fp) {
  if (!solver) return -1;
  if (t_solve) *t_solve = solver->t_solve_ms;
  if (t_bp) *t_bp = solver->t_backward_pass_ms;
  if (t_fp) *t_fp = solver->t_forward_pass_ms;
  return 0;
}

/* Cost-to-go recursion, k = N-1 .. 0:
 *   g   = P_{k+1} f_k + p_{k+1}
 *   Qx  = q_k + A' g            Qu  = r_k + B' g
 *   Qxx = Q_k + A' P_{k+1} A    Quu = R_k + B' P_{k+1} B    Qux = B' P_{k+1} A
 *   K_k = -Quu^-1 Qux           d_k = -Quu^-1 Qu
 *   P_k = Qxx + K' Quu K + K' Qux + Qux' K     p_k = Qx + K' Quu d + K' Qu + Qux' d            */
int ndlqr_BackwardPass(RiccatiSolver* solver) {
  if (!solver) return -1;
  const int N = solver->nhorizon;
  LQRData** knots = solver->prob->lqrdata;
  Matrix* Qx = &solver->Qx[0];   Matrix* g = &solver->Qx[1];
  Matrix* Qu = &solver->Qu[0];   Matrix* Quud = &solver->Qu[1];
  Matrix* Qxx = &solver->Qxx[0]; Matrix* AtP = &solver->Qxx[1];
  Matrix* Qux = &solver->Qux[0]; Matrix* BtP = &solver->Qux[1];
  Matrix* Quu = &solver->Quu[0]; Matrix* Lu = &solver->Quu[1];

  {
    Matrix Qn = ndlqr_GetQ(knots[N - 1]), qn = ndlqr_Getq(knots[N - 1]);
    MatrixCopyDiagonal(&solver->P[N - 1], &Qn);
    MatrixCopy(&solver->p[N - 1], &qn);
  }
  for (int k = N - 2; k >= 0; --k) {
    Matrix A = ndlqr_GetA(knots[k]), B = ndlqr_GetB(knots[k]), f = ndlqr_Getd(knots[k]);
    Matrix Qd = ndlqr_GetQ(knots[k]), q = ndlqr_Getq(knots[k]);
    Matrix Rd = ndlqr_GetR(knots[k]), r = ndlqr_Getr(knots[k]);
    Matrix* Pn = &solver->P[k + 1];
    Matrix* pn = &solver->p[k + 1];

    MatrixCopy(g, pn);
    MatrixMultiply(Pn, &f, g, 0, 0, 1.0, 1.0);
    MatrixMultiply(&A, g, Qx, 1, 0, 1.0, 0.0);
    MatrixMultiply(&B, g, Qu, 1, 0, 1.0, 0.0);
    MatrixAddition(&q, Qx, 1.0);
    MatrixAddition(&r, Qu, 1.0);

    MatrixCopyDiagonal(Qxx, &Qd);
    MatrixCopyDiagonal(Quu, &Rd);
    MatrixMultiply(&A, Pn, AtP, 1, 0, 1.0, 0.0);
    MatrixMultiply(&B, Pn, BtP, 1, 0, 1.0, 0.0);
    MatrixMultiply(AtP, &A, Qxx, 0, 0, 1.0, 1.0);
    MatrixMultiply(BtP, &B, Quu, 0, 0, 1.0, 1.0);
    MatrixMultiply(BtP, &A, Qux, 0, 0, 1.0, 0.0);

    Matrix* K = &solver->K[k];
    Matrix* d = &solver->d[k];
    MatrixCopy(Lu, Quu);
    MatrixCopy(K, Qux);
    MatrixCopy(d, Qu);
    CholeskyInfo info = DefaultCholeskyInfo();
    MatrixCholeskyFactorizeWithInfo(Lu, &info);
    MatrixCholeskySolveWithInfo(Lu, K, &info);
    MatrixCholeskySolveWithInfo(Lu, d, &info);
    FreeFactorization(&info);
    MatrixScaleByConst(K, -1.0);
    MatrixScaleByConst(d, -1.0);

    Matrix* P = &solver->P[k];
    Matrix* p = &solver->p[k];
    MatrixCopy(P, Qxx);
    MatrixMultiply(Quu, K, BtP, 0, 0, 1.0, 0.0);   /* Quu K (m x n scratch) */
    MatrixMultiply(K, BtP, P, 1, 0, 1.0, 1.0);
    MatrixMultiply(K, Qux, P, 1, 0, 1.0, 1.0);
    MatrixMultiply(Qux, K, P, 1, 0, 1.0, 1.0);
    MatrixCopy(p, Qx);
    MatrixMultiply(Quu, d, Quud, 0, 0, 1.0, 0.0);  /* Quu d */
    MatrixMultiply(K, Quud, p, 1, 0, 1.0, 1.0);
    MatrixMultiply(K, Qu, p, 1, 0, 1.0, 1.0);
    MatrixMultiply(Qux, d, p, 1, 0, 1.0, 1.0);
  }
  return 0;
}

/* Roll-out: y_k = P_k x_k + p_k, u_k = K_k x_k + d_k, x_{k+1} = A_k x_k + B_k u_k + f_k. */
int ndlqr_ForwardPass(RiccatiSolver* solver) {
  if (!solver) return -1;
  const int N = solver->nhorizon;
  LQRData** knots = solver->prob->lqrdata;
  Matrix x0 = {solver->nstates, 1, solver->prob->x0};
  MatrixCopy(&solver->X[0], &x0);
  for (int k = 0; k < N; ++k) {
    Matrix* x = &solver->X[k];
    MatrixCopy(&solver->Y[k], &solver->p[k]);
    MatrixMultiply(&solver->P[k], x, &solver->Y[k], 0, 0, 1.0, 1.0);
    if (k == N - 1) break;
    Matrix A = ndlqr_GetA(knots[k]), B = ndlqr_GetB(knots[k]), f = ndlqr_Getd(knots[k]);
    Matrix* u = &solver->U[k];
    Matrix* xn = &solver->X[k + 1];
    MatrixCopy(u, &solver->d[k]);
    MatrixMultiply(&solver->K[k], x, u, 0, 0, 1.0, 1.0);
    MatrixCopy(xn, &f);
    MatrixMultiply(&A, x, xn, 0, 0, 1.0, 1.0);
    MatrixMultiply(&B, u, xn, 0, 0, 1.0, 1.0);
  }
  return 0;
}

int ndlqr_SolveRiccati(RiccatiSolver* solver) {
  if (!solver) return -1;
  const clock_t t0 = clock();
  int err = ndlqr_BackwardPass(solver);
  const clock_t t1 = clock();
  if (!err) err = ndlqr_ForwardPass(solver);
  const clock_t t2 = clock();
  const double ms = 1000.0 / (double)CLOCKS_PER_SEC;
  solver->t_backward_pass_ms = (double)(t1 - t0) * ms;
  solver->t_forward_pass_ms = (double)(t2 - t1) * ms;
  solver->t_solve_ms = (double)(t2 - t0) * ms;
  return err;
}
