// kernels_mfma.hpp -- large-block Schur update on the fp64 matrix cores (gfx950).
//
// For block sizes that fill MFMA tiles (nstates a multiple of 16, 2*nstates + ninputs a multiple
// of 16 -- e.g. the (64,16) shape of BASELINE.json config 5) the Schur update
//     g(i, p) <- g(i, p) - E(i) * f_p            (ndlqr_UpdateShurFactor, nested_dissection.c:154-171)
// is a (2n+m) x n x n GEMM per knot and column: it runs on v_mfma_f64_16x16x4_f64 with f_p
// staged (negated) in LDS and E fragments in registers. Fast mode only: the MFMA accumulates its
// four products in its own order, so NDLQR_FLAG_STRICT_FP keeps the scalar schur_generic.
//
// Fragment layout of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3): lane l holds
// A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], and C/D[row = (l >> 4) + 4 * reg][col = l & 15].
#pragma once
#include "kernels_common.hpp"

namespace ndlqr {

typedef double mfma_acc_t __attribute__((ext_vector_type(4)));

// One workgroup (4 wavefronts) per knot; row tiles of 16 rows are dealt to the wavefronts.
//   NB = nstates / 16. grid (N, batch) -- or (2 * (N >> (l+1)), batch) in boundary mode --, block 256,
//   dynamic LDS = n * (n + 16) doubles.
template <int NB>
__global__ __launch_bounds__(256) void schur_mfma(Dims d, int l, double* F, double* z, int boundary,
                                                  const double* recs = nullptr) {
  constexpr int NX = 16 * NB, KS = NX / 4;
  // LDS row length with 2 * LDSW = 32 (mod 64) dwords: the two k-rows a 32-lane group reads hit disjoint banks
  constexpr int LDSW = (NX % 32 == 16) ? NX : NX + 16;
  extern __shared__ __attribute__((aligned(16))) double fl[];
  // boundary != 0: grid.x = 2 * (N >> (l+1)), first and last knot of every level-l subtree only
  const int N = d.N, rows = d.rows, b = blockIdx.y;
  const int i = boundary ? (blockIdx.x >> 1) * (2 << l) + ((blockIdx.x & 1) ? (2 << l) - 1 : 0) : blockIdx.x;
  const int half = 1 << l;
  const int base = (i >> (l + 1)) << (l + 1), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const bool left = i <= s;
  const bool calc_lambda = (i == 0) || (i & (half - 1)) != 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const double* E = Fblk(F, d, b, l, i);
  const int RT = rows / 16;

  for (int pass = 0; pass < 2; ++pass) {
    const int col = pass == 0 ? a : bb;
    if (col < 0) continue;  // uniform
    const bool created = pass == 0 ? !left : left;
    // stage -f_p (lambda rows of knot s+1 in column p), row-major [k][c], padded rows
    // recs (boundary pass of the record-based schedule): f_a / f_bb from the separator's record;
    // the lambda rows of the factor array are dead data there and are skipped below
    const double* f = recs ? recs + ((size_t)b * N + s) * (2 * (size_t)NX * NX + NX) + (pass == 0 ? 0 : NX * NX)
                           : Fblk(F, d, b, col, s + 1);
    __syncthreads();
    {  // all loads of the block before the first LDS store (a loop around load + store completes
       // them one after the other)
      constexpr int IT = NX * NX / 256;
      static_assert(NX * NX % 256 == 0, "whole rounds of the 256 threads");
      double t[IT];
#pragma unroll
      for (int it = 0; it < IT; ++it) t[it] = f[threadIdx.x + 256 * it];
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int e = threadIdx.x + 256 * it, k = e / NX, c = e - k * NX;
        fl[k * LDSW + c] = -t[it];
      }
    }
    __syncthreads();
    double* g = Fblk(F, d, b, col, i);
    for (int t = wave; t < RT; t += 4) {
      const bool lamtile = 16 * t < NX;  // NX % 16 == 0: a tile is entirely lambda rows or not
      if (lamtile && recs) continue;
      if (lamtile && !calc_lambda) {
        if (created && i != s + 1) {  // explicit zeros, like schur_generic
          for (int e = lane; e < 16 * NX; e += 64) g[(16 * t) * NX + e] = 0.0;
        }
        continue;
      }
      double afrag[KS];
      const double* Erow = E + (size_t)(16 * t + li) * NX + lk;
#pragma unroll
      for (int q = 0; q < KS; ++q) afrag[q] = Erow[4 * q];
      // the accumulators of all column tiles first (one load round), then the products
      mfma_acc_t accs[NB];
#pragma unroll
      for (int ct = 0; ct < NB; ++ct) {
        const double* gt = g + (size_t)(16 * t + lk) * NX + 16 * ct + li;  // row lk + 4 * reg
        if (created) { accs[ct][0] = 0.0; accs[ct][1] = 0.0; accs[ct][2] = 0.0; accs[ct][3] = 0.0; }
        else { accs[ct][0] = gt[0]; accs[ct][1] = gt[4 * NX]; accs[ct][2] = gt[8 * NX]; accs[ct][3] = gt[12 * NX]; }
      }
#pragma unroll
      for (int ct = 0; ct < NB; ++ct) {
        double* gt = g + (size_t)(16 * t + lk) * NX + 16 * ct + li;
        mfma_acc_t acc = accs[ct];
        const double* bcol = fl + lk * LDSW + 16 * ct + li;
#pragma unroll
        for (int q = 0; q < KS; ++q)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[q], bcol[4 * q * LDSW], acc, 0, 0, 0);
        gt[0] = acc[0]; gt[4 * NX] = acc[1]; gt[8 * NX] = acc[2]; gt[12 * NX] = acc[3];
      }
    }
  }
  // rhs entry per row: z(i)[r] -= E(r,:) . z_sep   (vector ALU; tiny next to the block products)
  const double* zsep = z + ((size_t)b * N + s + 1) * rows;
  for (int r = threadIdx.x; r < rows; r += 256) {
    if (r < NX && (recs || !calc_lambda)) continue;
    double* zp = z + ((size_t)b * N + i) * rows + r;
    const double* Erow = E + (size_t)r * NX;
    double acc = *zp;
    for (int k0 = 0; k0 < NX; k0 += 16) {  // sixteen operand pairs per load round, same order of summation
      double ev[16], zv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { ev[u] = Erow[k0 + u]; zv[u] = zsep[k0 + u]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fma(-ev[u], zv[u], acc);
    }
    *zp = acc;
  }
}

}  // namespace ndlqr
