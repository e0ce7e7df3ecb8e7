/*
 * linalg.c -- the reference's dense helper layer (src/linalg.c:22-232) routed to the GPU.
 *
 * The reference dispatches MatrixMultiply / MatrixCholesky* to one of {internal "clap", Eigen,
 * BLAS/LAPACKE}. Here there is exactly one backend: the HIP kernels behind ndlqr_hip_gemm /
 * _potrf_lower / _potrs_lower (host matrices are staged to HBM per call). These entry points
 * exist so that reference-style callers and the stage functions of stages.c keep working; the
 * solver hot path never goes through them. No CPU fallback: without a device they report
 * NDLQR_ERR_NO_DEVICE on stderr and leave the operands untouched.
 */
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "ndlqr.h"
#include "ndlqr_hip.h"

/* ---- global linear-algebra timer, same (non-reentrant) contract as src/linalg_utils.c ---- */
static struct {
  clock_t started;
  double total_ms;
} la_timer = {0, 0.0};

void MatrixLinAlgTimeStart(void) { la_timer.started = clock(); }
void MatrixLinAlgTimeStop(void) {
  la_timer.total_ms += (double)(clock() - la_timer.started) * 1000.0 / (double)CLOCKS_PER_SEC;
}
void MatrixLinAlgTimeReset(void) { la_timer.total_ms = 0.0; }
double MatrixGetLinAlgTimeMilliseconds(void) { return la_timer.total_ms; }

enum MatrixLinearAlgebraLibrary MatrixGetLinearAlgebraLibrary(void) { return libHIP; }

void MatrixPrintLinearAlgebraLibrary(void) {
  printf("Using HIP kernels on gfx950 (%d device(s) visible)\n", ndlqr_hip_device_count());
}

int MatrixAddition(Matrix* A, Matrix* B, double alpha) {
  if (!A || !B) return -1;
  /* B += alpha * A  ==  gemm with the identity is overkill; it is elementwise and tiny, but
   * arithmetic stays on the device for consistency: B = alpha * A * I(1x1 blocks) + 1 * B is
   * expressed as a (rows*cols x 1) * (1 x 1) product. */
  const int count = MatrixNumElements(A);
  double one = 1.0;
  return ndlqr_hip_gemm(0, 0, count, 1, 1, alpha, A->data, count, &one, 1, 1.0, B->data, count);
}

int MatrixCholeskyFactorize(Matrix* mat) {
  if (!mat) return -1;
  return ndlqr_hip_potrf_lower(mat->rows, mat->data, mat->rows);
}

int MatrixCholeskyFactorizeWithInfo(Matrix* mat, CholeskyInfo* cholinfo) {
  const int out = MatrixCholeskyFactorize(mat);
  if (cholinfo) {
    cholinfo->lib = 'H';
    cholinfo->is_freed = 1;
    cholinfo->success = out; /* 0 on success; the reference ends up storing `out` too (linalg.c:84) */
    cholinfo->uplo = 'L';
  }
  return out;
}

int MatrixCholeskySolve(Matrix* A, Matrix* b) {
  if (!A || !b) return -1;
  return ndlqr_hip_potrs_lower(A->rows, b->cols, A->data, A->rows, b->data, b->rows);
}

int MatrixCholeskySolveWithInfo(Matrix* A, Matrix* b, CholeskyInfo* cholinfo) {
  (void)cholinfo;
  return MatrixCholeskySolve(A, b);
}

void MatrixMultiply(Matrix* A, Matrix* B, Matrix* C, bool tA, bool tB, double alpha, double beta) {
  const int m = tA ? A->cols : A->rows;
  const int k = tA ? A->rows : A->cols;
  const int n = tB ? B->rows : B->cols;
  ndlqr_hip_gemm(tA, tB, m, n, k, alpha, A->data, A->rows, B->data, B->rows, beta, C->data, C->rows);
}

void MatrixSymmetricMultiply(Matrix* Asym, Matrix* B, Matrix* C, double alpha, double beta) {
  /* The reference reads only the lower triangle (linalg_custom.c:45-75). Mirror it into a
   * temporary full matrix on the host (pure data movement), multiply on the device. */
  const int n = Asym->rows;
  Matrix full = NewMatrix(n, n);
  for (int c = 0; c < n; ++c)
    for (int r = 0; r < n; ++r)
      full.data[r + n * c] = (r >= c) ? Asym->data[r + n * c] : Asym->data[c + n * r];
  MatrixMultiply(&full, B, C, false, false, alpha, beta);
  FreeMatrix(&full);
}

void MatrixCopyDiagonal(Matrix* dest, Matrix* src) {
  /* dest = diag(src): src is a vector of diagonal entries; everything else in dest is cleared
   * (src/linalg.c:215-221) */
  if (!dest || !src) return;
  const int n = dest->rows, len = src->rows * src->cols;
  memset(dest->data, 0, sizeof(double) * (size_t)dest->rows * dest->cols);
  for (int i = 0; i < len && i < n && i < dest->cols; ++i) dest->data[i + n * i] = src->data[i];
}

/* ---- the reference's internal-backend names (src/linalg_custom.c:6-138): thin aliases over the same
 *      device kernels, so that test/linalg_custom_test.c runs against this library unchanged ---- */
int clap_MatrixAddition(Matrix* A, Matrix* B, double alpha) { return MatrixAddition(A, B, alpha); }

int clap_MatrixScale(Matrix* A, double alpha) {
  if (!A) return -1;
  /* A <- 0 * (A * 1) + alpha * A as a (count x 1) product: the arithmetic stays on the device */
  const int count = MatrixNumElements(A);
  double one = 1.0;
  return ndlqr_hip_gemm(0, 0, count, 1, 1, 0.0, A->data, count, &one, 1, alpha, A->data, count);
}

int clap_MatrixMultiply(Matrix* A, Matrix* B, Matrix* C, bool tA, bool tB, double alpha, double beta) {
  if (!A || !B || !C) return -1;
  const int m = tA ? A->cols : A->rows;
  const int k = tA ? A->rows : A->cols;
  const int n = tB ? B->rows : B->cols;
  return ndlqr_hip_gemm(tA, tB, m, n, k, alpha, A->data, A->rows, B->data, B->rows, beta, C->data, C->rows);
}

int clap_MatrixTransposeMultiply(Matrix* A, Matrix* B, Matrix* C) {
  return clap_MatrixMultiply(A, B, C, true, false, 1.0, 0.0);
}

int clap_SymmetricMatrixMultiply(Matrix* Asym, Matrix* B, Matrix* C, double alpha, double beta) {
  if (!Asym || !B || !C) return -1;
  MatrixSymmetricMultiply(Asym, B, C, alpha, beta);
  return 0;
}

int clap_AddDiagonal(Matrix* A, double alpha) {
  if (!A) return -1;
  /* diag(A) += alpha: the n diagonal entries as a strided (1 x n) row, C <- alpha * 1 * ones + C */
  const int n = A->rows;
  Matrix ones = NewMatrix(1, n);
  MatrixSetConst(&ones, 1.0);
  double one = 1.0;
  const int err = ndlqr_hip_gemm(0, 0, 1, n, 1, alpha, &one, 1, ones.data, 1, 1.0, A->data, n + 1);
  FreeMatrix(&ones);
  return err;
}

int clap_CholeskyFactorize(Matrix* A) {
  const int out = MatrixCholeskyFactorize(A);
  return out == 0 ? clap_kCholeskySuccess : (out == -1 ? clap_kCholeskyFail : out);
}

int clap_CholeskySolve(Matrix* L, Matrix* b) { return MatrixCholeskySolve(L, b); }

int clap_LowerTriBackSub(Matrix* L, Matrix* b, bool istransposed) {
  if (!L || !b) return -1;
  return ndlqr_hip_trsv_lower(b->rows, b->cols, L->data, L->rows, b->data, b->rows, istransposed ? 1 : 0);
}
