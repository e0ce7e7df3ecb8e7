// kernels_bottom_reduced.hpp -- fast mode without KEEP: leaf phase + tree levels 0 and 1 on the
// REDUCED (separator-only) system, one wavefront per four consecutive knots, no knot states at all.
//
// Same mathematics as bottom_small<.., REDUCED> + reduced_level (kernels_small.hpp; DESIGN.md
// section 2): eliminating the states and inputs of every knot (ndlqr_SolveLeaf,
// src/nested_dissection.c:10-105) leaves a block-tridiagonal system in the multipliers,
//     S_s = [A_s | B_s] diag(1/Q_s, 1/R_s) [A_s | B_s]' + Q_{s+1}^-1          (diagonal block)
//     coupling of s to s-1:  -A_s Q_s^-1          coupling of s to s+1:  -Q_{s+1}^-1 A_{s+1}'
//     b~_s = [A_s | B_s] z(s).xu - z(s+1).lambda - z(s+1).x                     (leaf-phase rhs)
// and the nested-dissection levels (ndlqr_FactorInnerProduct, the separator Cholesky of
// src/solve.c:87-98, ndlqr_SolveCholeskyFactor, ndlqr_UpdateShurFactor) are block cyclic reduction
// on it. The wavefront of knots k0..k0+3 eliminates the level-0 separators s0 = k0 and s2 = k0+2
// straight from the problem data, hands their Schur contributions to the level-1 separator
// t = k0+1 in registers (matrix-core accumulator layout) and pushes what t and its children
// contribute to the two separators next to the group (k0-1, k0+3) into their RedSlot -- plain
// stores, one writer per element. Compared with the knot-based bottom kernel this drops the 28-row
// knot states, their Schur updates, the LDS publishing and every workgroup barrier: three
// factor_solve cores and seven small matrix-core products per wavefront.
#pragma once
#include "kernels_small.hpp"

namespace ndlqr {

typedef double acc4_t __attribute__((ext_vector_type(4)));

// Gram tiles of the forward-substituted panel Y = L^-1 [r_a | r_bb | b~] (see gram_mfma): the
// columns are re-filed as two 16-column tiles T0 = [a | z], T1 = [bb]; returns the requested ones of
//   g00 = T0'T0   g01 = T0'T1   g10 = T1'T0   g11 = T1'T1
// element (row lk + 4 g, column li) in component g of lane (li = lane & 15, lk = lane >> 4).
template <int NX, bool N00, bool N01, bool N10, bool N11>
__device__ __forceinline__ void gram_tiles(const int lane, double (&x)[NX], SepOut<NX>& out, acc4_t& g00,
                                           acc4_t& g01, acc4_t& g10, acc4_t& g11) {
  constexpr int LD = SepOut<NX>::LD, NC = SepOut<NX>::NC, KS = (NX + 3) / 4;
  static_assert(NC == 32 && NX + 1 <= 16, "two column tiles of 16: [a | z], [bb]");
  if (lane <= 2 * NX) {
    const int dst = lane < NX ? lane : (lane < 2 * NX ? 16 + (lane - NX) : NX);
#pragma unroll
    for (int k = 0; k < NX; ++k) out.X[k * LD + dst] = x[k];
  }
  wave_lds_sync();
  const int li = lane & 15, lk = lane >> 4;
  const acc4_t zero = {0.0, 0.0, 0.0, 0.0};
  g00 = zero; g01 = zero; g10 = zero; g11 = zero;
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int kk = 4 * q + lk, k = kk < NX ? kk : NX - 1;
    const double f0 = kk < NX ? out.X[k * LD + li] : 0.0;
    const double f1 = kk < NX ? out.X[k * LD + 16 + li] : 0.0;
    if constexpr (N00) g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, g00, 0, 0, 0);
    if constexpr (N01) g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f1, g01, 0, 0, 0);
    if constexpr (N10) g10 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f0, g10, 0, 0, 0);
    if constexpr (N11) g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, g11, 0, 0, 0);
  }
  wave_lds_sync();
}

//   grid (N / 4, batch), block 64; N >= 8; instances with matrix-core products only.
template <int NX, int NU>
__global__ __launch_bounds__(64, 4) void bottom_reduced(Dims d, const double* __restrict__ AB,
                                                        const double* __restrict__ QR,
                                                        const double* __restrict__ rhs, double* red,
                                                        double* __restrict__ rec, double* F,
                                                        int* __restrict__ info, const int store_l) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, LD = SepOut<NX>::LD, KS = (W + 3) / 4, SP = NX + 2;
  constexpr int QP = (NN + 63) / 64;
  static_assert(3 * NX <= 64, "three lane groups of NX store a record");
  __shared__ SepOut<NX> out[2];
  __shared__ double scr[NX * SP];   // S-bar from the accumulator layout to one row per lane
  __shared__ double rq[4 * W];      // 1 / [Q | R] of the four knots
  __shared__ double rh[4 * ROWS];   // their raw right-hand sides
  const int lane = threadIdx.x, b = blockIdx.y, N = d.N, k0 = blockIdx.x * 4;
  const int li = lane & 15, lk = lane >> 4;
  const int ri = li < NX ? li : NX - 1;  // rows / columns >= NX of a tile are padding: any finite data
  const bool hasA = k0 > 0, hasB = k0 + 4 < N;  // separators k0 - 1 / k0 + 3 exist
  const double* abm = AB + ((size_t)b * N + k0) * NX * W;  // [A | B] of the four knots, contiguous
  SEG_INIT();

  // ---- every global operand is requested up front
  {
    const double* q0 = QR + ((size_t)b * N + k0) * W;
    for (int e = lane; e < 4 * W; e += 64) {
      const double qv = q0[e];
      const int kn = e / W, c = e - kn * W;
      rq[e] = 1.0 / qv;
      if (!(qv > 0.0) && !(k0 + kn == N - 1 && c >= NX)) flag_failure(info, d, b);  // terminal R is unused
    }
    const double* r0 = rhs + ((size_t)b * N + k0) * ROWS;
    for (int e = lane; e < 4 * ROWS; e += 64) rh[e] = r0[e];
  }
  double pa0[QP], pb0[QP], pa2[QP], pb2[QP];  // A_s(i, j) and A_{s+1}(j, i) for s = s0, s2
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int e = lane + 64 * q, ec = e < NN ? e : NN - 1;
    const int i = ec / NX, j = ec - i * NX;
    pa0[q] = abm[i * W + j];
    pb0[q] = abm[NX * W + j * W + i];
    pa2[q] = abm[2 * NX * W + i * W + j];
    pb2[q] = abm[3 * NX * W + j * W + i];
  }
  double afr[3][KS];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int kq = 4 * q + lk, k = kq < W ? kq : W - 1;
      afr[kk][q] = abm[kk * NX * W + ri * W + k];
    }
  }
  wave_lds_sync();
  SEG(20);

  // [S-bar | b~] of separator k0 + kk from the problem data as ONE 16x16 tile (column NX of the B
  // operand carries the leaf-phase rhs of knot s). Knot 0 has its state fixed: its state columns
  // drop out of S-bar and carry x0 in the rhs (leaf phase of knot 0, src/nested_dissection.c:24-59).
  auto leaf_tile = [&](const int kk) -> acc4_t {
    const bool first = (k0 + kk == 0);
    const double* q0 = rq + kk * W;
    const double* q1 = rq + (kk + 1) * W;
    const double* z0 = rh + kk * ROWS;
    const double* z1 = rh + (kk + 1) * ROWS;
    acc4_t c;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = lk + 4 * g, ic = i < NX ? i : NX - 1;
      const double w1 = q1[ic];
      double v = 0.0;
      if (i < NX) {
        if (li < NX) v = (i == li) ? w1 : 0.0;
        else if (li == NX) v = -fma(z1[NX + ic], w1, z1[ic]);
      }
      c[g] = v;
    }
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int kq = 4 * q + lk, k = kq < W ? kq : W - 1;
      const bool kin = kq < W, fx = first && k < NX;
      const double av = afr[kk][q], wk = q0[k];
      const double zc = fx ? -z0[k] : z0[NX + k] * wk;
      const double a = kin ? av : 0.0;
      const double bv = !kin ? 0.0 : (li < NX ? (fx ? 0.0 : av * wk) : (li == NX ? zc : 0.0));
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, c, 0, 0, 0);
    }
    return c;
  };
  acc4_t c_s0 = leaf_tile(0), c_t = leaf_tile(1), c_s2 = leaf_tile(2);

  // panel columns [r_a | r_bb] of the two level-0 separators
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int e = lane + 64 * q;
    if (e < NN) {
      const int i = e / NX, j = e - i * NX;
      out[0].X[i * LD + j] = hasA ? -pa0[q] * rq[j] : 0.0;
      out[0].X[i * LD + NX + j] = -pb0[q] * rq[W + i];
      out[1].X[i * LD + j] = -pa2[q] * rq[2 * W + j];
      out[1].X[i * LD + NX + j] = hasB ? -pb2[q] * rq[3 * W + i] : 0.0;
    }
  }

  const int grp = lane / NX, gi = lane - grp * NX;
  // accumulator tile -> S-bar rows in registers (lanes of group 0) + rhs column of the panel
  auto tile_to_rows = [&](const acc4_t& c, SepOut<NX>& P, double (&acc)[NX]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = lk + 4 * g;
      if (i < NX) {
        if (li < NX) scr[i * SP + li] = c[g];
        else if (li == NX) P.X[i * LD + 2 * NX] = c[g];
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = scr[gi * SP + j];
  };
  auto store_record = [&](const int s, const bool ha, const bool hb, const SepOut<NX>& P) {
    double* myrec = rec + ((size_t)b * N + s) * (2 * NN + NX);
    if (grp < 2) {
      if (grp == 0 ? ha : hb) {
        double row[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) row[c] = P.X[gi * LD + grp * NX + c];
        store_row<NX>(myrec + grp * NN + gi * NX, row);
      }
    } else if (grp == 2) {
      myrec[2 * NN + gi] = P.X[gi * LD + 2 * NX];
    }
  };
  SEG(21);

  double acc[NX], Lrow[NX];
  acc4_t park_a, ca_t, unused;

  // ---- s0 = k0 (level 0, left child of t): DL[t], gL[t], CA[t]; its a-side faces separator k0 - 1
  tile_to_rows(c_s0, out[0], acc);
  if (factor_solve<NX, false, false, 16>(
          lane, acc, out[0], Lrow, store_l ? Fblk(F, d, b, 0, k0 + 1) : nullptr, [&](double (&x)[NX]) {
            acc4_t g11;
            gram_tiles<NX, true, false, true, true>(lane, x, out[0], park_a, unused, ca_t, g11);
#pragma unroll
            for (int g = 0; g < 4; ++g) c_t[g] -= (li < NX) ? g11[g] : ca_t[g];  // column NX: Y_bb' y_z
          }) &&
      lane == 0)
    flag_failure(info, d, b);
  store_record(k0, hasA, true, out[0]);
#pragma unroll
  for (int g = 0; g < 4; ++g) {  // r_a of t = -CA[t] = -Y_bb' Y_a
    const int i = lk + 4 * g;
    if (i < NX && li < NX) out[0].X[i * LD + li] = -ca_t[g];
  }

  // ---- s2 = k0 + 2 (level 0, right child of t): DR[t], gR[t], CB[t]; bb-side faces separator k0 + 3
  acc4_t park_b11, park_b01;
  tile_to_rows(c_s2, out[1], acc);
  if (factor_solve<NX, false, false, 16>(
            lane, acc, out[1], Lrow, store_l ? Fblk(F, d, b, 0, k0 + 3) : nullptr, [&](double (&x)[NX]) {
              acc4_t g00;
              gram_tiles<NX, true, true, false, true>(lane, x, out[1], g00, park_b01, unused, park_b11);
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int i = lk + 4 * g;
                if (li <= NX) c_t[g] -= g00[g];  // Y_a' [Y_a | y_z]
                if (i < NX && li < NX) out[0].X[i * LD + NX + li] = -park_b01[g];  // r_bb of t = -Y_a' Y_bb
              }
            }) &&
      lane == 0)
    flag_failure(info, d, b);
  store_record(k0 + 2, true, hasB, out[1]);

  // ---- t = k0 + 1 (level 1): pushes of the whole group to the separators k0 - 1 (A) and k0 + 3 (B)
  tile_to_rows(c_t, out[0], acc);
  const bool leftchild = (k0 & 4) == 0;
  const RedSlot<NX> sa = red_slot<NX>(red, d, b, hasA ? k0 - 1 : 3);
  const RedSlot<NX> sb = red_slot<NX>(red, d, b, hasB ? k0 + 3 : 3);
  if (factor_solve<NX, false, false, 16>(
            lane, acc, out[0], Lrow, store_l ? Fblk(F, d, b, 1, k0 + 2) : nullptr, [&](double (&x)[NX]) {
              acc4_t g00, g01, g11;
              gram_tiles<NX, true, true, false, true>(lane, x, out[0], g00, g01, unused, g11);
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int r = lk + 4 * g;
                if (hasA && r < NX && li <= NX) {
                  const double v = g00[g] + park_a[g];
                  if (li < NX) sa.DR()[r * NX + li] = v; else sa.gR()[r] = v;
                }
                if (hasB && li < NX) {
                  if (r < NX) {
                    sb.DL()[r * NX + li] = g11[g] + park_b11[g];
                    if (hasA) { if (leftchild) sb.CA()[li * NX + r] = g01[g]; else sa.CB()[r * NX + li] = g01[g]; }
                  } else if (r == NX) {
                    sb.gL()[li] = g01[g] + park_b01[g];
                  }
                }
              }
            }) &&
      lane == 0)
    flag_failure(info, d, b);
  store_record(k0 + 1, hasA, hasB, out[0]);
#ifdef NDLQR_SEGTIME
  __builtin_amdgcn_s_waitcnt(0);
#endif
  SEG(22);
}

}  // namespace ndlqr
