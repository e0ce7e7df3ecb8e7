// kernels_leaf.hpp -- the runtime-sized leaf kernel (ndlqr_SolveLeaf, nested_dissection.c:10-105),
// split from kernels_generic.hpp because the size-specialised launch sequence also uses it when
// the horizon is too short for bottom_small.
#pragma once
#include "kernels_common.hpp"

namespace ndlqr {

// ------------------------------------------------------------------------------------- leaves
// grid (N, batch), any block size. Writes both factor blocks of knot k completely (zero rows
// included) and the leaf-processed rhs block, so no memset is needed between solves.
template <bool STRICT>
__global__ void leaf_generic(Dims d, const double* __restrict__ AB, const double* __restrict__ QR,
                             const double* __restrict__ rhs, double* __restrict__ F,
                             double* __restrict__ z, int* __restrict__ info, const int lean = 0) {
  // lean (record-based schedule: fast mode without KEEP, runtime-sized path): the lambda rows of the
  // factor blocks are dead data there (see separator_generic) and are not written -- n of 2n + m rows
  const int k = blockIdx.x, b = blockIdx.y;
  const int n = d.n, m = d.m, w = d.w, rows = d.rows, N = d.N;
  const double* ab = AB + ((size_t)b * N + k) * n * w;
  const double* qr = QR + ((size_t)b * N + k) * w;
  const double* r0 = rhs + ((size_t)b * N + k) * rows;
  double* zk = z + ((size_t)b * N + k) * rows;
  const bool last = (k == N - 1);

  // pivot check (clap_CholeskyFactorize fails on a pivot <= 0, linalg_custom.c:99-102)
  for (int i = threadIdx.x; i < n + (last ? 0 : m); i += blockDim.x)
    if (!(qr[i] > 0.0)) flag_failure(info, d, b);

  // Fast mode: one reciprocal per weight instead of sqrt + two divisions per factor element
  // (L * L == q up to rounding; strict mode keeps the reference's operations)
  __shared__ double rqs[256];
  const bool use_rq = !STRICT && w <= 256;
  if (use_rq) {
    for (int i = threadIdx.x; i < w; i += blockDim.x) rqs[i] = 1.0 / qr[i];
    __syncthreads();
  }
  auto scaled = [&](const double x, const int i) -> double {  // x / Q_i (i < n) or x / R_{i-n}
    if (use_rq) return x * rqs[i];
    const double sq = qr[i] / sqrt(qr[i]);
    return (x / sq) / sq;
  };

  if (k == 0) {
    double* F0 = Fblk(F, d, b, 0, 0);
    for (int e = threadIdx.x + (lean ? n * n : 0); e < rows * n; e += blockDim.x) {
      const int r = e / n, c = e - r * n;
      double v = 0.0;
      if (r < n) {
        v = -ab[c * w + r];  // Fy = -A'
      } else if (r >= 2 * n) {
        const int i = r - 2 * n;
        v = scaled(ab[c * w + n + i], n + i);  // Fu = R \ B'
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
        v = scaled(r0[i], i - n);  // zu = R \ zu
      }
      zk[i] = v;
    }
    return;
  }

  const int lvl = trailing_ones(k), plvl = trailing_ones(k - 1);
  if (!last) {
    double* Fk = Fblk(F, d, b, lvl, k);
    for (int e = threadIdx.x + (lean ? n * n : 0); e < rows * n; e += blockDim.x) {
      const int r = e / n, c = e - r * n;
      double v = 0.0;
      if (r >= 2 * n) {
        const int i = r - 2 * n;
        v = scaled(ab[c * w + n + i], n + i);  // Fu = R \ B'
      } else if (r >= n) {
        const int i = r - n;
        v = scaled(ab[c * w + i], i);  // Fx = Q \ A'
      }
      Fk[e] = v;
    }
  }
  double* Fp = Fblk(F, d, b, plvl, k);
  for (int e = threadIdx.x + (lean ? n * n : 0); e < rows * n; e += blockDim.x) {
    const int r = e / n, c = e - r * n;
    double v = 0.0;
    if (r >= n && r < 2 * n && r - n == c) v = scaled(-1.0, c);  // Q \ (-I)
    Fp[e] = v;
  }
  for (int i = threadIdx.x; i < rows; i += blockDim.x) {
    double v = r0[i];
    if (i >= n && i < 2 * n) v = scaled(v, i - n);
    else if (i >= 2 * n && !last) v = scaled(v, i - n);
    zk[i] = v;
  }
}

}  // namespace ndlqr
