// kernels_generic.hpp -- runtime-sized kernels: correct for any (nstates, ninputs), used when no
// size-specialised instance exists (kernels_small.hpp) or when NDLQR_FLAG_GENERIC asks for a
// cross-check. One launch for the leaves, then two per tree level.
//
// What each kernel computes, and the reference code it stands for (under /root/reference/src):
//   leaf_generic       ndlqr_SolveLeaf                 nested_dissection.c:10-105
//   separator_generic  ndlqr_FactorInnerProduct        nested_dissection.c:114-134   (P1 / S1)
//                      Cholesky of S-bar               solve.c:87-98, linalg_custom.c:88-111 (P2)
//                      ndlqr_SolveCholeskyFactor       nested_dissection.c:136-152   (P3 / S2)
//   schur_generic      ndlqr_UpdateShurFactor          nested_dissection.c:154-177   (P4 / S3)
// Differences in organisation (not in arithmetic):
//   * the right-hand side rides along as one more column, so the reference's separate solution
//     sweep (solve.c:137-182) happens level by level together with the factorisation;
//   * only the structurally non-zero factor columns are touched: at level l a subtree owns
//     column l (being eliminated) and at most two outer columns a, bb (kernels_common.hpp);
//     every other column the reference multiplies through is exactly zero there;
//   * Q_k, R_k are diagonal by API contract (lqr_data.h:54-58), so their dense Cholesky
//     (linalg_custom.c:88-111) collapses to L_ii = Q_i / sqrt(Q_i) and the solves to two
//     divisions -- the very operations the reference executes on the non-zero entries.
#pragma once
#include "kernels_common.hpp"

namespace ndlqr {

// ------------------------------------------------------------------------------------- leaves
// grid (N, batch), any block size. Writes both factor blocks of knot k completely (zero rows
// included) and the leaf-processed rhs block, so no memset is needed between solves.
template <bool STRICT>
__global__ void leaf_generic(Dims d, const double* __restrict__ AB, const double* __restrict__ QR,
                             const double* __restrict__ rhs, double* __restrict__ F,
                             double* __restrict__ z, int* __restrict__ info) {
  const int k = blockIdx.x, b = blockIdx.y;
  const int n = d.n, m = d.m, w = d.w, rows = d.rows, N = d.N;
  const double* ab = AB + ((size_t)b * N + k) * n * w;
  const double* qr = QR + ((size_t)b * N + k) * w;
  const double* r0 = rhs + ((size_t)b * N + k) * rows;
  double* zk = z + ((size_t)b * N + k) * rows;
  const bool last = (k == N - 1);

  // pivot check (clap_CholeskyFactorize fails on a pivot <= 0, linalg_custom.c:99-102)
  for (int i = threadIdx.x; i < n + (last ? 0 : m); i += blockDim.x)
    if (!(qr[i] > 0.0)) atomicAdd(info + b, 1);

  if (k == 0) {
    double* F0 = Fblk(F, d, b, 0, 0);
    for (int e = threadIdx.x; e < rows * n; e += blockDim.x) {
      const int r = e / n, c = e - r * n;
      double v = 0.0;
      if (r < n) {
        v = -ab[c * w + r];  // Fy = -A'
      } else if (r >= 2 * n) {
        const int i = r - 2 * n;
        const double s = qr[n + i] / sqrt(qr[n + i]);
        v = (ab[c * w + n + i] / s) / s;  // Fu = R \ B'
      }
      F0[e] = v;
    }
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
      double v;
      if (i < n) {
        v = mad<STRICT>(-qr[i], r0[i], -r0[n + i]);  // zy = -Q*zy_old - zx_old
      } else if (i < 2 * n) {
        v = -r0[i - n];  // zx = -zy_old
      } else {
        const double s = qr[i - n] / sqrt(qr[i - n]);
        v = (r0[i] / s) / s;  // zu = R \ zu
      }
      zk[i] = v;
    }
    return;
  }

  const int lvl = trailing_ones(k), plvl = trailing_ones(k - 1);
  if (!last) {
    double* Fk = Fblk(F, d, b, lvl, k);
    for (int e = threadIdx.x; e < rows * n; e += blockDim.x) {
      const int r = e / n, c = e - r * n;
      double v = 0.0;
      if (r >= 2 * n) {
        const int i = r - 2 * n;
        const double s = qr[n + i] / sqrt(qr[n + i]);
        v = (ab[c * w + n + i] / s) / s;  // Fu = R \ B'
      } else if (r >= n) {
        const int i = r - n;
        const double s = qr[i] / sqrt(qr[i]);
        v = (ab[c * w + i] / s) / s;  // Fx = Q \ A'
      }
      Fk[e] = v;
    }
  }
  double* Fp = Fblk(F, d, b, plvl, k);
  for (int e = threadIdx.x; e < rows * n; e += blockDim.x) {
    const int r = e / n, c = e - r * n;
    double v = 0.0;
    if (r >= n && r < 2 * n && r - n == c) {
      const double s = qr[c] / sqrt(qr[c]);
      v = (-1.0 / s) / s;  // Q \ (-I)
    }
    Fp[e] = v;
  }
  for (int i = threadIdx.x; i < rows; i += blockDim.x) {
    double v = r0[i];
    if (i >= n && i < 2 * n) {
      const double s = qr[i - n] / sqrt(qr[i - n]);
      v = (v / s) / s;
    } else if (i >= 2 * n && !last) {
      const double s = qr[i - n] / sqrt(qr[i - n]);
      v = (v / s) / s;
    }
    zk[i] = v;
  }
}

// ------------------------------------------------------------------------------------- separators
// grid (N / 2^(l+1), batch), block 256, dynamic LDS = (3 n^2 + n) doubles.
// For the level-l separator s of each subtree: S-bar, the two outer right-hand sides fa, fb and
// the rhs vector; Cholesky; solves; results stored in the lambda rows of knot s+1.
template <bool STRICT>
__global__ void separator_generic(Dims d, int l, const double* __restrict__ AB, double* F, double* z,
                                  int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = d.n, m = d.m, w = d.w, N = d.N;
  const int b = blockIdx.y;
  const int half = 1 << l, base = blockIdx.x * (2 << l), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  double* S = sm;
  double* fa = S + n * n;
  double* fbm = fa + n * n;
  double* zs = fbm + n * n;

  const double* ab = AB + ((size_t)b * N + s) * n * w;
  const double* Es = Fblk(F, d, b, l, s);
  const double* Es1 = Fblk(F, d, b, l, s + 1);
  const double* Fas = a >= 0 ? Fblk(F, d, b, a, s) : nullptr;
  const double* Fbs1 = bb >= 0 ? Fblk(F, d, b, bb, s + 1) : nullptr;
  const double* zsl = z + ((size_t)b * N + s) * d.rows;
  double* zs1 = z + ((size_t)b * N + s + 1) * d.rows;

  // ---- P1: inner products. Row i of A_s, B_s against the state / input rows of knot s,
  //      minus the state rows of knot s+1 (its coupling block is [-I; 0]).
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int i = e / n, j = e - i * n;
    const double* arow = ab + i * w;
    double acc = 0.0;
    for (int k = 0; k < n; ++k) acc = mad<STRICT>(arow[k], Es[(n + k) * n + j], acc);
    for (int k = 0; k < m; ++k) acc = mad<STRICT>(arow[n + k], Es[(2 * n + k) * n + j], acc);
    S[e] = acc - Es1[(n + i) * n + j];
    if (a >= 0) {
      double acc2 = 0.0;
      for (int k = 0; k < n; ++k) acc2 = mad<STRICT>(arow[k], Fas[(n + k) * n + j], acc2);
      for (int k = 0; k < m; ++k) acc2 = mad<STRICT>(arow[n + k], Fas[(2 * n + k) * n + j], acc2);
      fa[e] = acc2;
    }
    if (bb >= 0) fbm[e] = -Fbs1[(n + i) * n + j];
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double* arow = ab + i * w;
    double acc = -zs1[i];  // beta = -1 on the old lambda entry (nested_dissection.c:125)
    for (int k = 0; k < n; ++k) acc = mad<STRICT>(arow[k], zsl[n + k], acc);
    for (int k = 0; k < m; ++k) acc = mad<STRICT>(arow[n + k], zsl[2 * n + k], acc);
    zs[i] = acc - zs1[n + i];
  }
  __syncthreads();

  // ---- P2: left-looking lower Cholesky in LDS, column by column.
  for (int j = 0; j < n; ++j) {
    for (int i = j + threadIdx.x; i < n; i += blockDim.x) {
      double acc = S[i * n + j];
      for (int k = 0; k < j; ++k) acc = mad<STRICT>(-S[i * n + k], S[j * n + k], acc);
      S[i * n + j] = acc;
    }
    __syncthreads();
    const double pivot = S[j * n + j];
    if (!(pivot > 0.0)) {  // uniform: every thread reads the same LDS word
      if (threadIdx.x == 0) atomicAdd(info + b, 1);
      break;
    }
    const double root = sqrt(pivot);
    __syncthreads();
    for (int i = j + threadIdx.x; i < n; i += blockDim.x) S[i * n + j] /= root;
    __syncthreads();
  }
  __syncthreads();

  // ---- P3: forward then transposed substitution, one thread per right-hand-side column.
  const int ncols = 2 * n + 1;
  for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
    double* x;
    int stride;
    if (c < n) { if (a < 0) continue; x = fa + c; stride = n; }
    else if (c < 2 * n) { if (bb < 0) continue; x = fbm + (c - n); stride = n; }
    else { x = zs; stride = 1; }
    for (int j = 0; j < n; ++j) {
      const double xj = x[j * stride] / S[j * n + j];
      x[j * stride] = xj;
      for (int i = j + 1; i < n; ++i) x[i * stride] = mad<STRICT>(-S[i * n + j], xj, x[i * stride]);
    }
    for (int j = n - 1; j >= 0; --j) {
      const double xj = x[j * stride] / S[j * n + j];
      x[j * stride] = xj;
      for (int i = 0; i < j; ++i) x[i * stride] = mad<STRICT>(-S[j * n + i], xj, x[i * stride]);
    }
  }
  __syncthreads();

  // ---- store into the lambda rows of knot s+1 (columns l, a, bb) and of the rhs
  double* outS = Fblk(F, d, b, l, s + 1);
  double* outa = a >= 0 ? Fblk(F, d, b, a, s + 1) : nullptr;
  double* outb = bb >= 0 ? Fblk(F, d, b, bb, s + 1) : nullptr;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    outS[e] = S[e];
    if (outa) outa[e] = fa[e];
    if (outb) outb[e] = fbm[e];
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) zs1[i] = zs[i];
}

// ------------------------------------------------------------------------------------- Schur update
// grid (ceil(N*rows*n / 256), batch), block 256. One thread per element (knot i, row r, col c):
//   g(i, p) -= E(i) * f_p   for the live outer columns p in {a, bb}, and the rhs (c == 0).
// Left-half knots own column a already (read-modify-write) and get column bb created
// (written, not accumulated); right-half knots the other way round.
template <bool STRICT>
__global__ void schur_generic(Dims d, int l, double* F, double* z) {
  const int n = d.n, rows = d.rows, N = d.N;
  const int b = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)N * rows * n) return;
  const int i = (int)(e / (rows * n));
  const int rem = (int)(e - (long)i * rows * n);
  const int r = rem / n, c = rem - r * n;
  const int half = 1 << l;
  const int base = (i >> (l + 1)) << (l + 1), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const bool left = i <= s;
  const bool calc_lambda = (i == 0) || (i & (half - 1)) != 0;  // nested_dissection.c:173-177
  if (r < n && !calc_lambda) {
    // lambda rows are not updated here; a freshly created block gets explicit zeros there,
    // except at knot s+1 whose lambda rows already hold f_a / f_bb from the separator kernel.
    if (i != s + 1) {
      if (a >= 0 && !left) Fblk(F, d, b, a, i)[r * n + c] = 0.0;
      if (bb >= 0 && left) Fblk(F, d, b, bb, i)[r * n + c] = 0.0;
    }
    return;
  }
  const double* Erow = Fblk(F, d, b, l, i) + r * n;
  if (a >= 0) {
    const double* f = Fblk(F, d, b, a, s + 1);  // lambda rows = f_a (n x n, row-major)
    double* g = Fblk(F, d, b, a, i) + r * n + c;
    double acc = left ? *g : 0.0;
    for (int k = 0; k < n; ++k) acc = mad<STRICT>(-Erow[k], f[k * n + c], acc);
    *g = acc;
  }
  if (bb >= 0) {
    const double* f = Fblk(F, d, b, bb, s + 1);
    double* g = Fblk(F, d, b, bb, i) + r * n + c;
    double acc = left ? 0.0 : *g;
    for (int k = 0; k < n; ++k) acc = mad<STRICT>(-Erow[k], f[k * n + c], acc);
    *g = acc;
  }
  if (c == 0) {
    const double* zsep = z + ((size_t)b * N + s + 1) * rows;
    double* g = z + ((size_t)b * N + i) * rows + r;
    double acc = *g;
    for (int k = 0; k < n; ++k) acc = mad<STRICT>(-Erow[k], zsep[k], acc);
    *g = acc;
  }
}

// ------------------------------------------------------------------------------------- dense helpers
// Device versions of the reference's internal routines, one element / column per thread.
// C = alpha*op(A)*op(B) + beta*C  (linalg_custom.c:20-43): beta first, then k ascending.
__global__ void dense_gemm(int tA, int tB, int m, int n, int k, double alpha, const double* A, int lda,
                           const double* B, int ldb, double beta, double* C, int ldc) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * n) return;
  const int i = e % m, j = e / m;
  double acc = C[i + (size_t)ldc * j] * beta;
  for (int p = 0; p < k; ++p) {
    const double a = tA ? A[p + (size_t)lda * i] : A[i + (size_t)lda * p];
    const double bv = tB ? B[j + (size_t)ldb * p] : B[p + (size_t)ldb * j];
    acc = acc + (alpha * a) * bv;
  }
  C[i + (size_t)ldc * j] = acc;
}

// in-place lower Cholesky (linalg_custom.c:88-111); single block.
__global__ void dense_potrf(int n, double* A, int lda, int* info) {
  for (int j = 0; j < n; ++j) {
    for (int i = j + threadIdx.x; i < n; i += blockDim.x) {
      double acc = A[i + (size_t)lda * j];
      for (int k = 0; k < j; ++k) acc = acc - A[i + (size_t)lda * k] * A[j + (size_t)lda * k];
      A[i + (size_t)lda * j] = acc;
    }
    __syncthreads();
    const double pivot = A[j + (size_t)lda * j];
    if (!(pivot > 0.0)) {
      if (threadIdx.x == 0) *info = j + 1;
      return;
    }
    const double root = sqrt(pivot);
    __syncthreads();
    for (int i = j + threadIdx.x; i < n; i += blockDim.x) A[i + (size_t)lda * j] /= root;
    __syncthreads();
  }
}

// L L' x = b (linalg_custom.c:113-138); one thread per right-hand side.
__global__ void dense_potrs(int n, int nrhs, const double* L, int ldl, double* B, int ldb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nrhs) return;
  double* x = B + (size_t)ldb * c;
  for (int j = 0; j < n; ++j) {
    const double xj = x[j] / L[j + (size_t)ldl * j];
    x[j] = xj;
    for (int i = j + 1; i < n; ++i) x[i] = x[i] - L[i + (size_t)ldl * j] * xj;
  }
  for (int j = n - 1; j >= 0; --j) {
    const double xj = x[j] / L[j + (size_t)ldl * j];
    x[j] = xj;
    for (int i = 0; i < j; ++i) x[i] = x[i] - L[j + (size_t)ldl * i] * xj;
  }
}

}  // namespace ndlqr
