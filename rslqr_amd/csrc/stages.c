/*
 * stages.c -- per-work-item stage functions on the solver's HOST mirrors.
 *
 * Same entry points, argument meaning and in-place effects as the reference's
 * src/nested_dissection.c (ndlqr_SolveLeaf :10-105, ndlqr_FactorInnerProduct :114-134,
 * ndlqr_SolveCholeskyFactor :136-152, ndlqr_UpdateShurFactor :154-171,
 * ndlqr_ShouldCalcLambda :173-177, ndlqr_ComputeShurCompliment :179-192), so the reference's
 * unit tests (test/nested_dissection_test.c) can be replayed against this library.
 *
 * All floating-point work goes through the device-backed Matrix* helpers (linalg.c ->
 * ndlqr_hip_gemm / potrf / potrs); only copies, negation-by-copy and zero fills happen on the
 * host. These functions are debugging / test surface: one H2D + kernel + D2H per dense call.
 * The production path (ndlqr_Solve, ndlqr_SolveBatch) never uses them.
 */
#include <stdio.h>
#include <string.h>

#include "ndlqr.h"

static NdFactor* block(NdData* nd, int index, int level) {
  NdFactor* out = NULL;
  ndlqr_GetNdFactor(nd, index, level, &out);
  return out;
}

static CholeskyInfo* q_info(NdLqrSolver* s, int k) {
  CholeskyInfo* out = NULL;
  ndlqr_GetQFactorizon(s->cholfacts, k, &out);
  return out;
}
static CholeskyInfo* r_info(NdLqrSolver* s, int k) {
  CholeskyInfo* out = NULL;
  ndlqr_GetRFactorizon(s->cholfacts, k, &out);
  return out;
}

/* dest = src, then dest <- chol \ dest */
static void copy_and_solve(Matrix* chol, CholeskyInfo* info, Matrix* dest, Matrix* src) {
  MatrixCopy(dest, src);
  MatrixCholeskySolveWithInfo(chol, dest, info);
}

static void leaf_first(NdLqrSolver* s) {
  const int n = s->nstates;
  NdFactor* C = block(s->data, 0, 0);
  NdFactor* F = block(s->fact, 0, 0);
  NdFactor* z = block(s->soln, 0, 0);
  Matrix* Q = &s->diagonals[0];
  Matrix* R = &s->diagonals[1];

  /* factor columns of the first knot: [Fy; Fx; Fu] = [-A'; 0; R \ B'] */
  MatrixCopy(&F->lambda, &C->state);
  MatrixScaleByConst(&F->lambda, -1.0);
  MatrixSetConst(&F->state, 0.0);
  MatrixCholeskyFactorizeWithInfo(R, r_info(s, 0));
  copy_and_solve(R, r_info(s, 0), &F->input, &C->input);
  MatrixCholeskySolveWithInfo(R, &z->input, r_info(s, 0));

  /* rhs of the first knot: zy <- -Q zy - zx, zx <- -zy(old). The reference parks zy(old) in
   * the unused lambda column of data(0,0) (nested_dissection.c:48-50); same here. */
  Matrix parked = {n, 1, C->lambda.data};
  MatrixCopy(&parked, &z->lambda);
  MatrixCopy(&z->lambda, &z->state);
  MatrixMultiply(Q, &parked, &z->lambda, false, false, -1.0, -1.0);
  MatrixCopy(&z->state, &parked);
  MatrixScaleByConst(&z->state, -1.0);
  MatrixCholeskyFactorizeWithInfo(Q, q_info(s, 0));
}

static void leaf_interior(NdLqrSolver* s, int k) {
  Matrix* Q = &s->diagonals[2 * k];
  CholeskyInfo* qi = q_info(s, k);
  NdFactor* z = block(s->soln, k, 0);
  MatrixCholeskyFactorizeWithInfo(Q, qi);

  if (k < s->nhorizon - 1) { /* everything that involves R_k, A_k, B_k */
    const int lvl = ndlqr_GetIndexLevel(&s->tree, k);
    NdFactor* C = block(s->data, k, lvl);
    NdFactor* F = block(s->fact, k, lvl);
    Matrix* R = &s->diagonals[2 * k + 1];
    CholeskyInfo* ri = r_info(s, k);
    MatrixCholeskyFactorizeWithInfo(R, ri);
    MatrixCholeskySolveWithInfo(R, &z->input, ri);
    copy_and_solve(Q, qi, &F->state, &C->state);
    copy_and_solve(R, ri, &F->input, &C->input);
  }
  MatrixCholeskySolveWithInfo(Q, &z->state, qi);

  /* coupling to the previous knot's dynamics: Q \ (-I), zero input part */
  const int plvl = ndlqr_GetIndexLevel(&s->tree, k - 1);
  NdFactor* Cp = block(s->data, k, plvl);
  NdFactor* Fp = block(s->fact, k, plvl);
  copy_and_solve(Q, qi, &Fp->state, &Cp->state);
  MatrixSetConst(&Fp->input, 0.0);
}

int ndlqr_SolveLeaf(NdLqrSolver* solver, int index) {
  if (!solver || index < 0 || index >= solver->nhorizon) return -1;
  if (index == 0) leaf_first(solver);
  else leaf_interior(solver, index);
  return 0;
}

int ndlqr_SolveLeaves(NdLqrSolver* solver) {
  if (!solver) return -1;
  for (int k = 0; k < solver->nhorizon; ++k) ndlqr_SolveLeaf(solver, k);
  return 0;
}

int ndlqr_FactorInnerProduct(NdData* data, NdData* fact, int index, int data_level, int fact_level) {
  NdFactor* C[2] = {block(data, index, data_level), block(data, index + 1, data_level)};
  NdFactor* F[2] = {block(fact, index, fact_level), block(fact, index + 1, fact_level)};
  if (!C[0] || !C[1] || !F[0] || !F[1]) return -1;
  Matrix S = F[1]->lambda;
  /* S <- C1x'F1x - S, then accumulate the other three products (beta = -1 only once) */
  double beta = -1.0;
  for (int side = 0; side < 2; ++side) {
    MatrixMultiply(&C[side]->state, &F[side]->state, &S, true, false, 1.0, beta);
    beta = 1.0;
    MatrixMultiply(&C[side]->input, &F[side]->input, &S, true, false, 1.0, beta);
  }
  return 0;
}

int ndlqr_SolveCholeskyFactor(NdData* fact, CholeskyInfo* cholinfo, int index, int level,
                              int upper_level) {
  if (!fact) return -1;
  if (upper_level <= level) fprintf(stderr, "ERROR: `upper_level` must be greater than `level`.");
  NdFactor* Sblock = block(fact, index + 1, level);
  NdFactor* fblock = block(fact, index + 1, upper_level);
  if (!Sblock || !fblock) return -1;
  MatrixCholeskySolveWithInfo(&Sblock->lambda, &fblock->lambda, cholinfo);
  return 0;
}

int ndlqr_UpdateShurFactor(NdData* fact, NdData* soln, int index, int i, int level,
                           int upper_level, bool calc_lambda) {
  if (!fact || !soln) return -1;
  NdFactor* fsep = block(soln, index + 1, upper_level);
  NdFactor* g = block(soln, i, upper_level);
  NdFactor* E = block(fact, i, level);
  if (!fsep || !g || !E) return -1;
  Matrix* f = &fsep->lambda;
  if (calc_lambda) MatrixMultiply(&E->lambda, f, &g->lambda, false, false, -1.0, 1.0);
  MatrixMultiply(&E->state, f, &g->state, false, false, -1.0, 1.0);
  MatrixMultiply(&E->input, f, &g->input, false, false, -1.0, 1.0);
  return 0;
}

bool ndlqr_ShouldCalcLambda(OrderedBinaryTree* tree, int index, int i) {
  const BinaryNode* sep = tree->node_list + index;
  if (i == 0) return true;
  return i != sep->left_inds.start && i != sep->right_inds.start;
}

int ndlqr_ComputeShurCompliment(NdLqrSolver* solver, int index, int level, int upper_level) {
  const BinaryNode* sep = solver->tree.node_list + index;
  NdData* target = (upper_level == 0) ? solver->soln : solver->fact;
  for (int i = sep->left_inds.start; i <= sep->right_inds.stop; ++i)
    ndlqr_UpdateShurFactor(solver->fact, target, index, i, level, upper_level,
                           ndlqr_ShouldCalcLambda(&solver->tree, index, i));
  return 0;
}
