// kernels_bottom_reduced.hpp -- fast mode without KEEP: the whole factorisation on the REDUCED
// (separator-only) system, no knot states at all. Default path of the batch API for the instances
// with matrix-core products (6 <= nstates <= 15).
//
//   bottom_reduced_mc   leaf phase + tree levels 0 and 1, one wavefront per four consecutive knots (bottom_group_mc)
//                       (TREE: the wavefront climbs on through the upper levels on arrival counters)
//   bottom8_reduced_mc  the same for eight knots per two-wavefront workgroup with tree level 2 behind it, the level-2
//                       slot in LDS (opt-in: NDLQR_FUSE2=1)
//   reduced_level_mc    one upper level, one wavefront per separator (reduced_separator_mc)
//   reduced_top_mc      the last three levels + the top-down sweep of the back-substitution, one workgroup per problem
//   the separator core  chol_pair_y_mc / chol_wy_mc (the Cholesky pass on the vector ALU, one row of S-bar per lane of
//                       every 16-lane DPP row; the panel -- and for separators that keep full records the unit
//                       vectors -- ride through it, one column per lane) + v_mfma_f64_16x16x4_f64 products chained
//                       through accumulator registers (tail_wy_mc, factor_tail_mc, gram_mc)
//
// eliminating the states and inputs of every knot (ndlqr_SolveLeaf,
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
// knot states, their Schur updates, the LDS publishing and every workgroup barrier.
#pragma once
#include "kernels_dpp.hpp"
#include "kernels_small.hpp"

namespace ndlqr {

typedef double acc4_t __attribute__((ext_vector_type(4)));

// ===================================================================================== matrix-core core
// The separator core of bottom_reduced_mc / reduced_level_mc. Only the Cholesky of S-bar and the forward substitutions
// that ride on it run on the vector ALU (rb_chol_inv, kernels_dpp.hpp: one row of S-bar per lane of every 16-lane DPP
// row, one right-hand-side column per lane); the rest is a chain of 16x16x4 matrix-core products through the
// accumulator registers. Component q of lane (li, lk) of a v_mfma_f64_16x16x4_f64 result is element (4 q + lk, li) --
// exactly the element that lane supplies in k-step q when the tile is the B operand of the next product and, read
// as the transposed tile, its A operand.
//
// Numerics (round 4). S-bar^-1 is never formed: with S-bar = L L', W = L^-1 and the solved panel
// Y = L^-1 [r_a | b~ | r_bb], the record is X = W'Y and everything pushed to other separators is a Gram product of Y.
// Round 3 formed S-bar^-1 = W'W and X = S-bar^-1 R; on problems with weak input costs (R ~ 1e-6: S-bar carries
// B R^-1 B' ~ 1e4 beside O(1) blocks) the rounding of the explicit inverse is not confined to the directions in which
// S-bar^-1 is small, and u = R^-1 (-r - B'y) amplified it to 5e-8 relative where the reference's substitutions
// give 5e-12 (tools/reduced_model.py reproduces both; tests: test_harder_families_large_and_padded_paths).
// Everything is written branch-free: loads are unconditional with clamped indices, the tiles are padded to 16 x 16
// in LDS (pad rows / columns of W are zero, so padding never reaches a result), conditions only select values.
template <int NX>
struct McPitch {
  static constexpr int SP = 18, WP = 17;  // row pitch of the S-bar tile / of W in LDS
};

// Tail of a separator whose W = L^-1 lies in LDS (Wm: row pitch McPitch::WP, zero outside the leading NX x NX block)
// and whose panel is still raw: c = the separator's [S-bar | b~] tile (only its rhs column NX is read here), ra / rb =
// B-operand fragments of r_a / r_bb, i.e. ra[q] = r_a(4 q + lk, li) (any finite value outside the block).
//   Y = W [r_a | b~ | r_bb]     X = W'Y
// On return X0 = [f_a | z_sep], X1 = [f_bb] in accumulator layout. hook(Y0, Y1, Y0, Y1) sees the solved panel in both
// operand roles (the fragment arrays and the accumulator tiles hold the same numbers): Gram products Y'Y.
// (The level-0 separators of the schedules that keep full records: tree schedule, KEEP_RECORDS.)
template <int NX, class Hook>
__device__ __forceinline__ void factor_tail_mc(const int lane, const acc4_t& c, const double (&ra)[(NX + 3) / 4],
                                               const double (&rb)[(NX + 3) / 4], const double* Wm, acc4_t& X0,
                                               acc4_t& X1, Hook hook) {
  constexpr int KS = (NX + 3) / 4, WP = McPitch<NX>::WP;
  int lane_o = lane;  // (opaque: the lane predicates of one core are recomputed, not kept in scalar registers)
  asm volatile("" : "+v"(lane_o));
  const int li = lane_o & 15, lk = lane_o >> 4;
  double wt[KS], wa[KS], b0[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    wt[q] = Wm[(4 * q + lk) * WP + li];  // W(k, li): A operand of W'(i, k)
    wa[q] = Wm[li * WP + 4 * q + lk];    // W(li, k): A operand of W(i, k)
    // columns beyond the blocks carry finite don't-cares: they only reach tile elements nobody reads
    b0[q] = li == NX ? c[q] : ra[q];
  }
  const acc4_t zero = {0.0, 0.0, 0.0, 0.0};
  acc4_t Y0 = zero, Y1 = zero;
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    Y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[q], b0[q], Y0, 0, 0, 0);
    Y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[q], rb[q], Y1, 0, 0, 0);
  }
  X0 = zero; X1 = zero;
  double y0[KS], y1[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    y0[q] = Y0[q]; y1[q] = Y1[q];
    X0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[q], Y0[q], X0, 0, 0, 0);
    X1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[q], Y1[q], X1, 0, 0, 0);
  }
  hook(y0, y1, Y0, Y1);
}

// The pass of ONE separator whose record keeps X = S-bar^-1 R (every level >= 1): the paired pass (chol_pair_y_mc
// below) with both halves on the SAME tile. DPP rows 0-1 (lanes 0..31) carry the panel, one column per lane --
// h = lane & 31: h < NX column h of r_a, h == NX the rhs column b~ (taken from the tile here), NX < h <= 2 NX column
// h - NX - 1 of r_bb -- and leave the pass with Y = L^-1 [r_a | b~ | r_bb]; rows 2-3 carry the unit vectors and leave
// with W = L^-1. The forward substitution of a column costs the same FMAs whatever the column is, and the recurrence
// is per 16-lane row: with one separator per wavefront the rows would otherwise repeat each other's work.
//   w: in = this lane's panel column (lanes < 32 other than the b~ lane; the rest: anything finite), clobbered.
// LDS: the S-bar tile at buf[0, 16 SP); then, over it, the tiles Y0 = L^-1 [r_a | b~], Y1 = L^-1 r_bb (rows = k,
// pitch YP, 4 KS rows) and W (pitch WP, 16 rows; zero outside the leading NX x NX block).
template <int NX>
struct McWyLayout {
  static constexpr int SP = McPitch<NX>::SP, WP = McPitch<NX>::WP, KS = (NX + 3) / 4, YP = 17, TILE = 4 * KS * YP;
  static constexpr int Y0 = 0, Y1 = TILE, W = 2 * TILE;
  static constexpr int SIZE = 2 * TILE + 16 * WP > 16 * SP ? 2 * TILE + 16 * WP : 16 * SP;
  static_assert(2 * NX + 1 <= 32, "the panel fits the lanes of two DPP rows");
};
template <int NX>
__device__ __forceinline__ bool chol_wy_mc(const int lane_in, const acc4_t& c, double* buf, double (&w)[NX],
                                           double* lstore) {
  using P = McWyLayout<NX>;
  constexpr int SP = P::SP, YP = P::YP, KS = P::KS;
  static_assert(P::YP == P::WP, "one store stride for both kinds of lanes");
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  const int li = lane & 15, lk = lane >> 4, h = lane & 31;
  const int ri = li < NX ? li : NX - 1;  // lanes NX..15 of a 16-lane row are padding: they repeat row NX - 1
#pragma unroll
  for (int g = 0; g < 4; ++g) buf[(lk + 4 * g) * SP + li] = c[g];
  wave_lds_sync();
  double acc[NX];
  if constexpr (NX % 2 == 0) {
#pragma unroll
    for (int j = 0; j < NX; j += 2) {
      const double2 t = *reinterpret_cast<const double2*>(&buf[ri * SP + j]);
      acc[j] = t.x; acc[j + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = buf[ri * SP + j];
  }
  if (lk >= 2) {  // (plain selects / loads under lane predicates: no cross-lane operation inside)
#pragma unroll
    for (int k = 0; k < NX; ++k) w[k] = (k == li) ? 1.0 : 0.0;
  } else if (h == NX) {
#pragma unroll
    for (int k = 0; k < NX; ++k) w[k] = buf[k * SP + NX];
  }
  const bool bad = rb_chol_inv<NX, false>(li, acc, w);
  wave_lds_sync();  // (every lane has its row of S-bar and the rhs column: the tiles may overwrite the S-bar tile)
  {
    // rows 0-1: column h of Y0 (h <= NX), column h - NX - 1 of Y1 (h <= 2 NX), the idle lanes dump into the pad column
    // of Y1; rows 2-3: column li of W (twice the same values: benign duplicates; lanes li >= NX hold zeros).
    // Unconditional stores: a store under a lane predicate makes the compiler sink the recurrence behind it.
    const int ycol = h <= NX ? P::Y0 + h : P::Y1 + (h <= 2 * NX ? h - NX - 1 : YP - 1);
    double* dst = buf + (lk < 2 ? ycol : P::W + li);
#pragma unroll
    for (int k = 0; k < NX; ++k) dst[k * YP] = w[k];
    // rows NX .. 4 KS - 1 of the Y tiles (read by the last k-step) and rows NX .. 15 of W: zero
    if constexpr (4 * KS > NX) {
#pragma unroll
      for (int e0 = 0; e0 < (4 * KS - NX) * YP; e0 += 32) {
        const int e = e0 + h;
        if (e < (4 * KS - NX) * YP && lk < 2) {
          buf[P::Y0 + NX * YP + e] = 0.0;
          buf[P::Y1 + NX * YP + e] = 0.0;
        }
      }
    }
#pragma unroll
    for (int e0 = 0; e0 < (16 - NX) * P::WP; e0 += 32) {
      const int e = e0 + h;
      if (e < (16 - NX) * P::WP && lk >= 2) buf[P::W + NX * P::WP + e] = 0.0;
    }
  }
  if (lstore && lane < NX) store_row<NX>(lstore + li * NX, acc);
  wave_lds_sync();
  return bad;
}

// ... and its matrix-core tail: X = W'Y (the record), hook(Y0, Y1, Y0, Y1) for the Gram products.
template <int NX, class Hook>
__device__ __forceinline__ void tail_wy_mc(const int lane, const double* buf, acc4_t& X0, acc4_t& X1, Hook hook) {
  using P = McWyLayout<NX>;
  constexpr int KS = P::KS;
  int lane_o = lane;
  asm volatile("" : "+v"(lane_o));
  const int li = lane_o & 15, lk = lane_o >> 4;
  double wt[KS], y0[KS], y1[KS];
  acc4_t Z0 = {0.0, 0.0, 0.0, 0.0}, Z1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    wt[q] = buf[P::W + (4 * q + lk) * P::WP + li];  // W(k, li): A operand of W'(i, k)
    y0[q] = buf[P::Y0 + (4 * q + lk) * P::YP + li];
    y1[q] = buf[P::Y1 + (4 * q + lk) * P::YP + li];
    Z0[q] = y0[q]; Z1[q] = y1[q];
  }
  const acc4_t zero = {0.0, 0.0, 0.0, 0.0};
  X0 = zero; X1 = zero;
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    X0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[q], y0[q], X0, 0, 0, 0);
    X1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[q], y1[q], X1, 0, 0, 0);
  }
  hook(y0, y1, Z0, Z1);
}

// The Cholesky + inverse of TWO independent separators in one pass: DPP rows 0-1 (lanes 0..31) work on tile cA,
// rows 2-3 on cB -- the row-broadcast recurrence (rb_chol_inv) is per 16-lane row, and with one separator all four
// rows repeat the same work. bottom_reduced_mc eliminates its two level-0 separators this way: two passes per
// four-knot group instead of three (132 DPP FMAs, 12 pivots and their hazard fences per wavefront less).
// LDS: buf[0, 288) S-bar tile A, [288, 576) tile B, [576, 848) W_A; W_B lies over tile A (read before it is written).
// Returns per lane: the separator of this lane's DPP row had a non-positive pivot.
template <int NX>
struct McPairLayout {
  static constexpr int SP = McPitch<NX>::SP, WP = McPitch<NX>::WP;
  static constexpr int SCR_A = 0, SCR_B = 16 * SP, W_A = 32 * SP, W_B = 0, SIZE = 32 * SP + 16 * WP;
  static_assert(16 * WP <= 16 * SP, "W_B fits over the S-bar tile of A");
};
template <int NX>
__device__ __forceinline__ bool chol_pair_mc(const int lane_in, const acc4_t& cA, const acc4_t& cB, double* buf,
                                             double* lstoreA, double* lstoreB) {
  using P = McPairLayout<NX>;
  constexpr int SP = P::SP, WP = P::WP;
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  const int li = lane & 15, lk = lane >> 4;
  const int ri = li < NX ? li : NX - 1;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    buf[P::SCR_A + (lk + 4 * g) * SP + li] = cA[g];
    buf[P::SCR_B + (lk + 4 * g) * SP + li] = cB[g];
  }
  wave_lds_sync();
  const double* mine = buf + (lk < 2 ? P::SCR_A : P::SCR_B) + ri * SP;
  double acc[NX], w[NX];
  if constexpr (NX % 2 == 0) {
#pragma unroll
    for (int j = 0; j < NX; j += 2) {
      const double2 t = *reinterpret_cast<const double2*>(&mine[j]);
      acc[j] = t.x; acc[j + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = mine[j];
  }
  const bool bad = rb_chol_inv<NX>(li, acc, w);
  wave_lds_sync();  // (every lane has its row of S-bar: W_B may overwrite tile A)
  {  // rows 0-1 store W_A, rows 2-3 W_B, every lane its column (two identical copies each: benign duplicates;
     // unconditional stores, see chol_wy_mc)
    double* wdst = buf + (lk < 2 ? P::W_A : P::W_B) + li;
#pragma unroll
    for (int r = 0; r < NX; ++r) wdst[r * WP] = w[r];
    // rows NX..15 of both: zero (the tiles that lay here were not)
#pragma unroll
    for (int e0 = 0; e0 < (16 - NX) * WP; e0 += 32) {
      const int e = e0 + (lane & 31);
      if (e < (16 - NX) * WP) buf[(lk < 2 ? P::W_A : P::W_B) + NX * WP + e] = 0.0;
    }
  }
  if (lstoreA && lane < NX) store_row<NX>(lstoreA + li * NX, acc);
  if (lstoreB && lane >= 32 && lane < 32 + NX) store_row<NX>(lstoreB + li * NX, acc);
  wave_lds_sync();
  return bad;
}

// The pass for two separators whose records are COMPACT (level 0 of the default schedule): instead of W = L^-1 (the
// forward substitution of the unit vectors) the lanes carry the panel itself, one column each -- half h = lane & 31 of
// DPP rows 0-1 (separator A) / 2-3 (B): h < NX column h of r_a, h == NX the rhs column b~, NX < h <= 2 NX column
// h - NX - 1 of r_bb -- and leave the pass with Y = L^-1 [r_a | b~ | r_bb]: the six matrix-core products Y = W R of a
// level-0 separator and the LDS round trip of W disappear, the Gram hooks read Y from LDS tiles (Y0 = L^-1 [r_a | b~],
// Y1 = L^-1 r_bb, rows = k, pitch 17), and the compact record keeps L itself (packed lower triangle, row i by lane i;
// rb_backsub substitutes with it).  w: in = this lane's panel column (the b~ lanes: anything), out = its column of Y.
// LDS: S-bar tiles A | B at buf[0, 576); then the four Y tiles of 4 KS rows over them.
template <int NX>
struct McPairYLayout {
  static constexpr int SP = McPitch<NX>::SP, KS = (NX + 3) / 4, YP = 17, TILE = 4 * KS * YP;
  static constexpr int SCR_A = 0, SCR_B = 16 * SP;
  static constexpr int SIZE = 4 * TILE > 32 * SP ? 4 * TILE : 32 * SP;
  // tile t of separator x (x: 0 = A, 1 = B; t: 0 = Y0, 1 = Y1)
  static constexpr int tile(const int x, const int t) { return (2 * x + t) * TILE; }
};
template <int NX>
__device__ __forceinline__ bool chol_pair_y_mc(const int lane_in, const acc4_t& cA, const acc4_t& cB, double* buf,
                                               double (&w)[NX], double* lrecA, double* lrecB) {
  using P = McPairYLayout<NX>;
  constexpr int SP = P::SP, YP = P::YP, KS = P::KS;
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  const int li = lane & 15, lk = lane >> 4, h = lane & 31;
  const int ri = li < NX ? li : NX - 1;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    buf[P::SCR_A + (lk + 4 * g) * SP + li] = cA[g];
    buf[P::SCR_B + (lk + 4 * g) * SP + li] = cB[g];
  }
  wave_lds_sync();
  const double* tile = buf + (lk < 2 ? P::SCR_A : P::SCR_B);
  double acc[NX];
  if constexpr (NX % 2 == 0) {
#pragma unroll
    for (int j = 0; j < NX; j += 2) {
      const double2 t = *reinterpret_cast<const double2*>(&tile[ri * SP + j]);
      acc[j] = t.x; acc[j + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = tile[ri * SP + j];
  }
  if (h == NX) {  // the rhs column of the tile (plain loads under a lane predicate: no cross-lane operation inside)
#pragma unroll
    for (int k = 0; k < NX; ++k) w[k] = tile[k * SP + NX];
  }
  const bool bad = rb_chol_inv<NX, false>(li, acc, w);
  wave_lds_sync();  // (every lane has its row of S-bar and the rhs column: the Y tiles may overwrite the S-bar tiles)
  {
    const int x = lk >> 1;
    // column h of Y0 (h <= NX), column h - NX - 1 of Y1 (h <= 2 NX); the idle lanes dump into the pad column of Y1
    double* ydst = buf + (h <= NX ? P::tile(x, 0) + h : P::tile(x, 1) + (h <= 2 * NX ? h - NX - 1 : YP - 1));
#pragma unroll
    for (int k = 0; k < NX; ++k) ydst[k * YP] = w[k];
    if constexpr (4 * KS > NX) {  // rows NX .. 4 KS - 1 of the tiles are read by the last k-step: zero
#pragma unroll
      for (int e0 = 0; e0 < (4 * KS - NX) * YP; e0 += 16) {
        const int e = e0 + li;
        if (e < (4 * KS - NX) * YP) buf[P::tile(x, lk & 1) + NX * YP + e] = 0.0;
      }
    }
  }
  // the record: row li of L (entries 0 .. li) by lane li of DPP rows 0 (A) and 2 (B)
  if ((lk & 1) == 0 && li < NX) {
    double* lr = (lk == 0 ? lrecA : lrecB) + li * (li + 1) / 2;
#pragma unroll
    for (int c = 0; c < NX; ++c)
      if (c <= li) lr[c] = acc[c];
  }
  wave_lds_sync();
  return bad;
}

// Schur-complement blocks R' S-bar^-1 R = R'X of the panel R = [R0 | R1] (operand fragments, the
// same registers that fed X = S-bar^-1 R) and X = [X0 | X1] (accumulator layout = B operand):
//   g00 = R0'X0   g01 = R0'X1   g10 = R1'X0   g11 = R1'X1
template <int NX, bool N00, bool N01, bool N10, bool N11>
__device__ __forceinline__ void gram_mc(const double (&R0)[(NX + 3) / 4], const double (&R1)[(NX + 3) / 4],
                                        const acc4_t& X0, const acc4_t& X1, acc4_t& g00, acc4_t& g01, acc4_t& g10,
                                        acc4_t& g11) {
  constexpr int KS = (NX + 3) / 4;
  const acc4_t zero = {0.0, 0.0, 0.0, 0.0};
  g00 = zero; g01 = zero; g10 = zero; g11 = zero;
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    if constexpr (N00) g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(R0[q], X0[q], g00, 0, 0, 0);
    if constexpr (N01) g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(R0[q], X1[q], g01, 0, 0, 0);
    if constexpr (N10) g10 = __builtin_amdgcn_mfma_f64_16x16x4f64(R1[q], X0[q], g10, 0, 0, 0);
    if constexpr (N11) g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(R1[q], X1[q], g11, 0, 0, 0);
  }
}

// Component g of an accumulator tile holds row lk + 4 g: rows_all(g) -- the row is inside the
// block for every lane; rows_none(g) -- for no lane (both known at compile time, so that the
// stores below sit in as few predicated regions as possible).
template <int NX> __device__ __forceinline__ constexpr bool rows_all(int g) { return 4 * g + 3 < NX; }
template <int NX> __device__ __forceinline__ constexpr bool rows_none(int g) { return 4 * g >= NX; }

// record f_a | f_bb | z_sep of separator s from the solved tiles
template <int NX>
__device__ __forceinline__ void store_record_mc(double* __restrict__ myrec, const int lane, const bool ha,
                                                const bool hb, const acc4_t& X0, const acc4_t& X1) {
  constexpr int NN = NX * NX;
  const int li = lane & 15, lk = lane >> 4;
  if (li < NX) {
    if (ha) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = lk + 4 * g;
        if (!rows_none<NX>(g) && (rows_all<NX>(g) || r < NX)) myrec[r * NX + li] = X0[g];
      }
    }
    if (hb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = lk + 4 * g;
        if (!rows_none<NX>(g) && (rows_all<NX>(g) || r < NX)) myrec[NN + r * NX + li] = X1[g];
      }
    }
  } else if (li == NX) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int r = lk + 4 * g;
      if (!rows_none<NX>(g) && (rows_all<NX>(g) || r < NX)) myrec[2 * NN + r] = X0[g];
    }
  }
}

// What a separator contributes to the reduced system of its two neighbours (see reduced_level),
// from its Gram tiles g00 = Y_a'[Y_a | y_z], g01 = [Y_a | y_z]'Y_bb, g11 = Y_bb'Y_bb plus whatever
// its children parked (pa, pb01, pb11; zero tiles for none):
//   A: DR += g00 (columns < NX), gR += g00 (column NX)      B: DL += g11, gL += g01 (row NX)
//   coupling of the parent to the other neighbour: CA[B] = g01' (left child) or CB[A] = g01.
// put(p, v): store (first writer of the launch) or atomic add; set(p, v): store of a single-writer
// block (the couplings). Under the tree schedule the stores are write-through (agent-scope
// relaxed atomic stores, `sc1`): the reader may sit on another XCD, whose L2 is not coherent with
// the writer's.
struct StorePlain { __device__ __forceinline__ void operator()(double* p, double v) const { *p = v; } };
struct StoreThrough {
  __device__ __forceinline__ void operator()(double* p, double v) const {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};
// (Round 3 tried plain read-modify-writes in the level-per-launch schedule -- every accumulator block has exactly one
//  writer per launch -- with the reads requested in the wavefront's first round of loads: same kernel time, 17 % more
//  HBM traffic in the upper levels than the memory-side atomics, which read nothing. Dropped: DESIGN.md section 7.)
struct AddAtomic { __device__ __forceinline__ void operator()(double* p, double v) const { atomicAdd(p, v); } };

template <int NX, class Put, class Set>
__device__ __forceinline__ void push_mc(const int lane, const bool hasA, const bool hasB, const bool leftchild,
                                        const RedSlot<NX>& sa, const RedSlot<NX>& sb, const acc4_t& g00,
                                        const acc4_t& g01, const acc4_t& g11, const acc4_t& pa, const acc4_t& pb01,
                                        const acc4_t& pb11, Put put, Set set) {
  const int li = lane & 15, lk = lane >> 4;
  if (hasA && li <= NX) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int r = lk + 4 * g;
      // (the lower triangle of the symmetric block alone, packed; column NX: gR)
      if (!rows_none<NX>(g) && (rows_all<NX>(g) || r < NX) && (li <= r || li == NX))
        put(li < NX ? sa.DR() + r * (r + 1) / 2 + li : sa.gR() + r, g00[g] + pa[g]);
    }
  }
  if (hasB && li < NX) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int r = lk + 4 * g;
      if (!rows_none<NX>(g) && (rows_all<NX>(g) || r < NX) && li <= r) put(sb.DL() + r * (r + 1) / 2 + li, g11[g] + pb11[g]);
    }
    if (lk == NX % 4) put(sb.gL() + li, g01[NX / 4] + pb01[NX / 4]);  // row NX of g01: y_z' Y_bb
    if (hasA) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r = lk + 4 * g;
        if (!rows_none<NX>(g) && (rows_all<NX>(g) || r < NX))
          set(leftchild ? sb.CA() + li * NX + r : sa.CB() + r * NX + li, g01[g]);
      }
    }
  }
}

// [S-bar | b~] of a separator from the problem data staged in LDS, as ONE 16x16 tile: column NX of
// the B operand carries the leaf-phase rhs of knot s.
//   am: [A_s | B_s] (row pitch WP), q0: 1 / [Q_s | R_s], q1: 1 / Q_{s+1}, z0: rhs(s), z1: rhs(s+1).
// first: s == 0 -- knot 0 has its state fixed: its state columns drop out of S-bar and carry x0 in
// the rhs (leaf phase of knot 0, src/nested_dissection.c:24-59). init(g): what else goes into
// element (lk + 4 g, li) (the pushed blocks of an upper level).
// NEXT = false: the terms of knot s + 1 (Q_{s+1}^-1 on the diagonal, z(s+1) in the rhs column) are not staged in LDS: q1 /
// z1 are not read, init(g) supplies them (bottom8_reduced_mc: the level-2 separator's next knot belongs to the other
// wavefront of the workgroup).
template <int NX, int NU, int WP, bool NEXT = true, class Init>
__device__ __forceinline__ acc4_t leaf_tile_mc(const int lane, const bool first, const double* am, const double* q0,
                                               const double* q1, const double* z0, const double* z1, Init init) {
  constexpr int W = NX + NU, KS = (W + 3) / 4;
  const int li = lane & 15, lk = lane >> 4;
  const int ri = li < NX ? li : NX - 1;  // rows / columns >= NX of a tile are padding: any finite data
  // every LDS operand first (one wait for all of them), then the arithmetic
  double w1[4], za[4], zb[4], in0[4], av[KS], wk[KS], zraw[KS];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int i = lk + 4 * g, ic = i < NX ? i : NX - 1;
    if constexpr (NEXT) { w1[g] = q1[ic]; za[g] = z1[ic]; zb[g] = z1[NX + ic]; }
    else { w1[g] = 0.0; za[g] = 0.0; zb[g] = 0.0; }
    in0[g] = init(g);
  }
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int kq = 4 * q + lk, k = kq < W ? kq : W - 1;
    const bool fx = first && k < NX;
    av[q] = am[ri * WP + k]; wk[q] = q0[k]; zraw[q] = z0[fx ? k : NX + k];
  }
  acc4_t c;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int i = lk + 4 * g;
    // rows / columns beyond the block: finite don't-cares
    c[g] = (li == NX ? -fma(zb[g], w1[g], za[g]) : ((i == li) ? w1[g] : 0.0)) + in0[g];
  }
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int kq = 4 * q + lk, k = kq < W ? kq : W - 1;
    const bool kin = kq < W, fx = first && k < NX;
    const double sv = fx ? 0.0 : av[q] * wk[q];            // S-bar columns
    const double zc = fx ? -zraw[q] : zraw[q] * wk[q];     // rhs column
    const double bsel = li == NX ? zc : sv;
    if constexpr (W % 4 == 0) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bsel, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f64_16x16x4f64(kin ? av[q] : 0.0, bsel, c, 0, 0, 0);
  }
  return c;
}

// store_l == 2 (NDLQR_FLAG_KEEP_RECORDS on the default schedule, round 4): the Cholesky factor of a separator of level >= 1
// -- n x n, row-major, what the record-based re-solve substitutes with (rb_forward, kernels_rowbcast.hpp) -- goes into the
// slack of the record slot in front of it: slot s - 1 belongs to a level-0 separator, whose compact record uses
// n (n + 1) / 2 of the slot's 2 n^2 + n doubles. (store_l == 1: into the factor array, the full-record schedules.)
template <int NX>
__device__ __forceinline__ constexpr int rb_lrec_offset() { return (NX * (NX + 1) / 2 + 1) / 2 * 2; }
// where the Cholesky factor of the separator s of level >= 1 (odd s) is kept: n x n, row-major
template <int NX>
__device__ __forceinline__ double* rb_lrec(double* rec, const Dims& d, const int b, const int s) {
  return rec + ((size_t)b * d.N + (s - 1)) * (2 * NX * NX + NX) + rb_lrec_offset<NX>();
}
template <int NX>
__device__ __forceinline__ const double* rb_lrec(const double* rec, const Dims& d, const int b, const int s) {
  return rec + ((size_t)b * d.N + (s - 1)) * (2 * NX * NX + NX) + rb_lrec_offset<NX>();
}

template <int NX>
__device__ __forceinline__ double* lstore_of(const int store_l, double* F, double* rec, const Dims& d, const int b,
                                             const int level, const int s) {
  return store_l == 2 ? rb_lrec<NX>(rec, d, b, s) : (store_l ? Fblk(F, d, b, level, s + 1) : nullptr);
}

// LDS of one wavefront working on the reduced system: a buffer that first holds the staged inputs
// and then, once the tiles and the panel columns are in registers, the tiles of the pass; plus the
// reciprocal weights and right-hand sides of the knots involved. BOTTOM: sized for the four knots of the
// bottom levels (whose wavefront may go on to upper levels under the tree schedule: the larger of the two);
// else for slot + one knot of an upper level alone (reduced_level_mc, reduced_top_mc: 6.8 instead of 8.1 KB
// at (12,4), so that the LDS no longer caps those launches at 19 wavefronts per CU).
template <int NX, int NU, bool BOTTOM = true>
struct alignas(16) ReducedLds {
  static constexpr int W = NX + NU, ROWS = 2 * NX + NU, WP = (W % 2 == 0) ? W + 2 : W;
  static constexpr int SLOT = RedSlot<NX>::SIZE, NSC = McWyLayout<NX>::SIZE;  // the pass of a separator of level >= 1
  static constexpr int NB0 = BOTTOM ? 4 * NX * WP : 0, NB1 = SLOT + NX * WP;
  static constexpr int NB2 = BOTTOM ? McPairLayout<NX>::SIZE : 0;   // the paired Cholesky of the bottom levels
  static constexpr int NB3 = BOTTOM ? McPairYLayout<NX>::SIZE : 0;  // ... and its compact-record form (Y tiles)
  static constexpr int NB01 = (NB0 > NB1 ? NB0 : NB1) > NSC ? (NB0 > NB1 ? NB0 : NB1) : NSC;
  static constexpr int NB23 = NB2 > NB3 ? NB2 : NB3;
  static constexpr int NBUF = NB01 > NB23 ? NB01 : NB23;
  static constexpr int NRQ = BOTTOM ? 4 * W : (W + NX + 1) / 2 * 2, NRH = BOTTOM ? 4 * ROWS : (NX + W + 2 * NX + 1) / 2 * 2;
  double buf[NBUF];
  double rq[NRQ];
  double rh[NRH];
#ifdef NDLQR_DEV_LDS_PAD  // developer builds: fewer wavefronts per CU (occupancy experiments)
  double dev_pad[BOTTOM ? NDLQR_DEV_LDS_PAD / 8 : 1];
#endif
};

// One separator of an upper level (l >= 2) of the separator-only schedule on the matrix-core core
// (see reduced_level): subtree [base, base + 2^(l+1)) of problem b, by one wavefront.
// TREE: called from the tree schedule -- the slot was written by wavefronts of this launch, possibly
// on another XCD: it is read with L1-bypassing (`sc1`) loads and the couplings are stored through.
// EXT: the separator's slot is not in `red` but at `slot_ext` in LDS, complete (bottom8_reduced_mc: its children pushed
// there); nothing of it is loaded from memory.
template <int NX, int NU, bool TREE, class LdsT, bool EXT = false>
__device__ __forceinline__ void reduced_separator_mc(const Dims& d, const int l, const int base, const int b,
                                                     const int lane, const double* __restrict__ AB,
                                                     const double* __restrict__ QR,
                                                     const double* __restrict__ rhs, double* red,
                                                     double* __restrict__ rec, double* F, int* __restrict__ info,
                                                     const int store_l, LdsT& lds, const double* slot_ext = nullptr) {
  constexpr int W = NX + NU, NN = NX * NX, KSN = (NX + 3) / 4;
  constexpr int WP = ReducedLds<NX, NU>::WP, SLOT = RedSlot<NX>::SIZE;
  // slot and [A | B] are dead once the tile and the panel columns are in registers: the tiles of the
  // pass (McWyLayout) lie over them
  const double* slot = EXT ? slot_ext : lds.buf;  // DL | DR | CA | CB | gL | gR of this separator
  double* abs_ = lds.buf + SLOT;   // [A_s | B_s]
  double* rq = lds.rq;             // 1 / [Q_s | R_s], 1 / Q_{s+1}
  double* zs = lds.rh;             // rhs(s), rhs(s+1).lambda | x
  static_assert(LdsT::NRQ >= W + NX && LdsT::NRH >= NX + W + 2 * NX, "the shared arrays are large enough for a level");
  const int N = d.N;
  const int T = 2 << l, s = base + (1 << l) - 1;
  const bool hasA = base > 0, hasB = base + T < N;
  const int li = lane & 15, lk = lane >> 4;
  const int ri = li < NX ? li : NX - 1;
  SEG_INIT();
  const bool leftchild = (base & T) == 0;
  const RedSlot<NX> sa = red_slot<NX>(red, d, b, hasA ? base - 1 : s);
  const RedSlot<NX> sb = red_slot<NX>(red, d, b, hasB ? base + T - 1 : s);

  // ---- ONE round of coalesced loads into LDS (the slot is contiguous and 16-byte aligned); the
  //      operand fragments are gathered from there. (Knots s and s+1 of a level >= 2 are never the
  //      first or the last knot: no special cases.)
  {
    const double* sl = red_slot<NX>(red, d, b, s).p;
    const double* abm = AB + ((size_t)b * N + s) * NX * W;
    const double* qr = QR + ((size_t)b * N + s) * W;             // knot s: W entries, then the Q of knot s + 1
    const double* r0 = rhs + ((size_t)b * N + s) * (2 * NX + NU);  // rhs(s) | rhs(s+1).lambda | rhs(s+1).x
    constexpr int NS = SLOT / 2, IS = (NS + 63) / 64, NA = NX * W, IA = (NA + 63) / 64;
    constexpr int NQ = W + NX, IQ = (NQ + 63) / 64, NR = NX + W + 2 * NX, IR = (NR + 63) / 64;
    static_assert(SLOT % 2 == 0, "slot copied as 16-byte words");
    double2 ts[IS];
    double ta[IA], tq[IQ], tr[IR];
#pragma unroll
    for (int it = 0; it < (EXT ? 0 : IS); ++it) {
      const int e = lane + 64 * it, ec = e < NS ? e : NS - 1;
      if constexpr (TREE) {
        ts[it].x = __hip_atomic_load(sl + 2 * ec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ts[it].y = __hip_atomic_load(sl + 2 * ec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        ts[it] = reinterpret_cast<const double2*>(sl)[ec];
      }
    }
#pragma unroll
    for (int it = 0; it < IA; ++it) { const int e = lane + 64 * it; ta[it] = abm[e < NA ? e : NA - 1]; }
#pragma unroll
    for (int it = 0; it < IQ; ++it) { const int e = lane + 64 * it; tq[it] = qr[e < NQ ? e : NQ - 1]; }
#pragma unroll
    for (int it = 0; it < IR; ++it) { const int e = lane + 64 * it; tr[it] = r0[e < NR ? e : NR - 1]; }
    // The stores are unconditional, on the same clamped index as the loads (surplus lanes rewrite the
    // last element with its own value): a store under a lane predicate makes the compiler sink the
    // load behind the predicate too, and the loads then complete one after the other.
#pragma unroll
    for (int it = 0; it < (EXT ? 0 : IS); ++it) {
      const int e = lane + 64 * it, ec = e < NS ? e : NS - 1;
      reinterpret_cast<double2*>(lds.buf)[ec] = ts[it];
    }
#pragma unroll
    for (int it = 0; it < IA; ++it) {
      const int e = lane + 64 * it, ec = e < NA ? e : NA - 1, row = ec / W, c = ec - row * W;
      abs_[row * WP + c] = ta[it];
    }
#pragma unroll
    for (int it = 0; it < IQ; ++it) { const int e = lane + 64 * it, ec = e < NQ ? e : NQ - 1; rq[ec] = 1.0 / tq[it]; }
#pragma unroll
    for (int it = 0; it < IR; ++it) { const int e = lane + 64 * it, ec = e < NR ? e : NR - 1; zs[ec] = tr[it]; }
  }
  wave_lds_sync();
  constexpr int TRI = RedSlot<NX>::TRI;
  const double *DL = slot, *DR = slot + TRI, *CA = slot + 2 * TRI, *CB = slot + 2 * TRI + NN;
  const double *gL = slot + 2 * TRI + 2 * NN, *gR = slot + 2 * TRI + 2 * NN + NX;

  // [S-bar | b~] = leaf tile - DL - DR | - gL - gR
  const acc4_t c0 = leaf_tile_mc<NX, NU, WP>(lane, false, abs_, rq, rq + W, zs, zs + NX + W, [&](int g) {
    const int i = lk + 4 * g, ic = i < NX ? i : NX - 1;
    const int tix = RedSlot<NX>::tri(ic, ri);
    const double dd = DL[tix] + DR[tix], gg = gL[ic] + gR[ic];
    return -(li == NX ? gg : dd);
  });
  // the panel, one column per lane of DPP rows 0-1 (chol_wy_mc): r_a = -CA, r_bb = -CB, straight from the staged slot
  double wcol[NX];
  {
    const int h = lane & 31;
    const bool is_a = h < NX, is_b = h > NX && h <= 2 * NX;
    const double* src = is_a ? CA + h : CB + (is_b ? h - NX - 1 : 0);
    const double sign = ((is_a && hasA) || (is_b && hasB)) ? -1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) wcol[k] = src[k * NX] * sign;
  }
  wave_lds_sync();  // last read of the staged operands
  SEG(9);

  acc4_t X0, X1, unused;
  if (chol_wy_mc<NX>(lane, c0, lds.buf, wcol, lstore_of<NX>(store_l, F, rec, d, b, l, s)) && lane == 0)
    flag_failure(info, d, b);
  SEG(31);
  tail_wy_mc<NX>(lane, lds.buf, X0, X1,
                 [&](const double (&R0)[KSN], const double (&R1)[KSN], const acc4_t& Z0, const acc4_t& Z1) {
                   acc4_t g00, g01, g11;
                   gram_mc<NX, true, true, false, true>(R0, R1, Z0, Z1, g00, g01, unused, g11);
                   const acc4_t zero = {0.0, 0.0, 0.0, 0.0};
                   if constexpr (TREE)
                     push_mc<NX>(lane, hasA, hasB, leftchild, sa, sb, g00, g01, g11, zero, zero, zero, AddAtomic(),
                                 StoreThrough());
                   else
                     push_mc<NX>(lane, hasA, hasB, leftchild, sa, sb, g00, g01, g11, zero, zero, zero, AddAtomic(),
                                 StorePlain());
                 });
  SEG(13);
  store_record_mc<NX>(rec + ((size_t)b * N + s) * (2 * NN + NX), lane, hasA, hasB, X0, X1);
#ifdef NDLQR_SEGTIME
  __builtin_amdgcn_s_waitcnt(0);
#endif
  SEG(12);
}

//   grid (N >> (l+1), batch), block 64; l >= 2.
template <int NX, int NU>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void reduced_level_mc(
    Dims d, int l, const double* __restrict__ AB, const double* __restrict__ QR, const double* __restrict__ rhs,
    double* red, double* __restrict__ rec, double* F, int* __restrict__ info, const int store_l) {
  __shared__ ReducedLds<NX, NU, false> lds;
  reduced_separator_mc<NX, NU, false>(d, l, (blockIdx.x + d.xoff) * (2 << l), blockIdx.y, threadIdx.x, AB, QR, rhs, red, rec, F,
                                      info, store_l, lds);
}


// The level-2 separator s = base + 3 of bottom8_reduced_mc from its leaf tile (registers) and its slot (LDS): what
// reduced_separator_mc does behind its loads and its leaf tile, in two parts with a workgroup barrier between them.
// Part 1 (first wavefront): S-bar assembly and the pass (chol_wy_mc) -> the tiles Y0, Y1, W in `buf`.
template <int NX, int NU>
__device__ __forceinline__ void reduced_eliminate_pass_mc(const Dims& d, const int base, const int b, const int lane,
                                                          const acc4_t& c_leaf, const double* slot,
                                                          int* __restrict__ info, double* buf) {
  constexpr int NN = NX * NX, TRI = RedSlot<NX>::TRI;
  const int N = d.N, T = 8;
  const bool hasA = base > 0, hasB = base + T < N;
  const int li = lane & 15, lk = lane >> 4, ri = li < NX ? li : NX - 1;
  const double *DL = slot, *DR = slot + TRI, *CA = slot + 2 * TRI, *CB = slot + 2 * TRI + NN;
  const double *gL = slot + 2 * TRI + 2 * NN, *gR = slot + 2 * TRI + 2 * NN + NX;
  acc4_t c0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int i = lk + 4 * g, ic = i < NX ? i : NX - 1;
    const int tix = RedSlot<NX>::tri(ic, ri);
    const double dd = DL[tix] + DR[tix], gg = gL[ic] + gR[ic];
    c0[g] = c_leaf[g] - (li == NX ? gg : dd);
  }
  double wcol[NX];
  {
    const int h = lane & 31;
    const bool is_a = h < NX, is_b = h > NX && h <= 2 * NX;
    const double* src = is_a ? CA + h : CB + (is_b ? h - NX - 1 : 0);
    const double sign = ((is_a && hasA) || (is_b && hasB)) ? -1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) wcol[k] = src[k * NX] * sign;
  }
  wave_lds_sync();
  if (chol_wy_mc<NX>(lane, c0, buf, wcol, nullptr) && lane == 0) flag_failure(info, d, b);
}
// Part 2, shared by the two wavefronts: `gram` -- the Gram products of Y and the pushes (atomic adds behind the groups'
// plain stores); else -- X = W'Y and the record.
template <int NX, int NU>
__device__ __forceinline__ void reduced_eliminate_tail_mc(const Dims& d, const int base, const int b, const int lane,
                                                          const bool gram, double* red, double* __restrict__ rec,
                                                          const double* buf) {
  using P = McWyLayout<NX>;
  constexpr int NN = NX * NX, KS = P::KS;
  const int N = d.N, T = 8, s = base + 3;
  const bool hasA = base > 0, hasB = base + T < N, leftchild = (base & T) == 0;
  const int li = lane & 15, lk = lane >> 4;
  double y0[KS], y1[KS];
  acc4_t Z0 = {0.0, 0.0, 0.0, 0.0}, Z1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    y0[q] = buf[P::Y0 + (4 * q + lk) * P::YP + li];
    y1[q] = buf[P::Y1 + (4 * q + lk) * P::YP + li];
    Z0[q] = y0[q]; Z1[q] = y1[q];
  }
  if (gram) {  // (uniform per wavefront)
    const RedSlot<NX> sa = red_slot<NX>(red, d, b, hasA ? base - 1 : s);
    const RedSlot<NX> sb = red_slot<NX>(red, d, b, hasB ? base + T - 1 : s);
    acc4_t g00, g01, g11, unused;
    gram_mc<NX, true, true, false, true>(y0, y1, Z0, Z1, g00, g01, unused, g11);
    const acc4_t zero = {0.0, 0.0, 0.0, 0.0};
    push_mc<NX>(lane, hasA, hasB, leftchild, sa, sb, g00, g01, g11, zero, zero, zero, AddAtomic(), StorePlain());
  } else {
    double wt[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) wt[q] = buf[P::W + (4 * q + lk) * P::WP + li];
    acc4_t X0 = {0.0, 0.0, 0.0, 0.0}, X1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      X0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[q], y0[q], X0, 0, 0, 0);
      X1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wt[q], y1[q], X1, 0, 0, 0);
    }
    store_record_mc<NX>(rec + ((size_t)b * N + s) * (2 * NN + NX), lane, hasA, hasB, X0, X1);
  }
}

// Leaf phase + tree levels 0, 1 AND 2 in one launch: a workgroup of two wavefronts per EIGHT knots (round 4). Each wavefront
// runs the four-knot group of bottom_reduced_mc; what the two groups push to the level-2 separator m = k0 + 3 between
// them -- DL | gL | CA from the left group, DR | gR | CB from the right one -- goes to a slot in LDS instead of `red`, and
// behind a workgroup barrier the first wavefront takes m through the pass, from there and from the leaf tile it formed
// while [A_m | B_m] was staged (nothing is loaded from memory behind the barrier: the other wavefront's slot idles
// meanwhile); behind a second barrier the two wavefronts share the tail (Gram products + pushes / X = W'Y + record).
// Gone: the level-2 launch, the slot of every level-2 separator (written by pushes, read back: 480 + 324 doubles per eight knots) and its
// second read of [A | B] are gone; m's own pushes to the separators k0 - 1 and k0 + 7 follow the groups' plain stores
// as atomic adds (the groups' stores are acknowledged before the barrier), like those of a level launch.
//   grid (N / 8, batch), block 128; N >= 16 (a level-3 separator exists); compact level-0 records.
template <int NX, int NU, bool TREE, bool REDIRECT>
__device__ __forceinline__ void bottom_group_mc(const Dims& d, const int k0, const int b, const int lane,
                                                const double* __restrict__ AB, const double* __restrict__ QR,
                                                const double* __restrict__ rhs, double* red, double* __restrict__ rec,
                                                double* F, int* __restrict__ info, const int store_l, const int compact0,
                                                ReducedLds<NX, NU>& lds, double* slotA, double* slotB,
                                                const bool with_m = false, acc4_t* c_m = nullptr);  // (below)
template <int NX, int NU>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4))) void bottom8_reduced_mc(
    Dims d, const double* __restrict__ AB, const double* __restrict__ QR, const double* __restrict__ rhs, double* red,
    double* __restrict__ rec, int* __restrict__ info) {
  __shared__ ReducedLds<NX, NU> lds[2];
  __shared__ __attribute__((aligned(16))) double mslot[RedSlot<NX>::SIZE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), b = blockIdx.y;
  const int kw = (blockIdx.x + d.xoff) * 8;
  int lane = threadIdx.x & 63;
  // (parts of the slot that no group writes -- the coupling to a neighbour that does not exist -- are read and
  //  multiplied by zero: they have to be finite)
  for (int e = threadIdx.x; e < RedSlot<NX>::SIZE; e += 128) mslot[e] = 0.0;
  __syncthreads();
  acc4_t c_m = {0.0, 0.0, 0.0, 0.0};  // leaf tile of m (the first wavefront: it has [A_m | B_m] staged)
  bottom_group_mc<NX, NU, false, true>(d, kw + 4 * wave, b, lane, AB, QR, rhs, red, rec, nullptr, info, 0, 1, lds[wave],
                                       wave == 1 ? mslot : nullptr, wave == 0 ? mslot : nullptr, wave == 0, &c_m);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's plain stores to k0 - 1 / k0 + 7 are acknowledged
  __syncthreads();
  asm volatile("" : "+v"(lane));
  if (wave == 0) reduced_eliminate_pass_mc<NX, NU>(d, kw, b, lane, c_m, mslot, info, lds[0].buf);
  __syncthreads();
  // (the tail is shared: the first wavefront forms the Gram products and pushes, the second X = W'Y and the record)
  reduced_eliminate_tail_mc<NX, NU>(d, kw, b, lane, wave == 0, red, rec, lds[0].buf);
}

// The top of the tree in ONE launch: the last three levels (4 + 2 + 1 separators per problem), one workgroup of four
// wavefronts per problem, one separator per wavefront and level, a workgroup barrier between levels. These levels
// are bound by the latency of a single wavefront (~4.5 us per separator however few there are), and as launches of
// their own each paid that latency plus a launch boundary: 14 + 9 + 6.5 us at (12,4,256) x 1024. The hand-off between
// levels goes through global memory like the tree schedule's (write-through pushes, L1-bypassing slot loads,
// reduced_separator_mc<TREE>): the four wavefronts share a CU, but nothing is assumed about its L1.
//   grid (batch), block 256; l0 = K - 3 >= 2.
// ytop != nullptr: the workgroup goes straight on to the top-down sweep over the records of the separators of level
// >= 3 of its problem (backsub_top_body, kernels_rowbcast.hpp; its [N / 8][NX] array lies over the four scratches:
// N / 8 * NX doubles <= 4 * sizeof(ReducedLds) is checked on the host) -- one launch and its ~15 us less.
template <int NX>
__device__ __forceinline__ void backsub_top_body(const Dims& d, const int b, const int t,
                                                 const double* __restrict__ recs, double* __restrict__ ytop,
                                                 double* ytop_lds, const int bp = -1, const double* zsep = nullptr);
template <int NX, int NU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void reduced_top_mc(Dims d, const int l0, const double* __restrict__ AB,
                                                      const double* __restrict__ QR, const double* __restrict__ rhs,
                                                      double* red, double* __restrict__ rec, double* F,
                                                      int* __restrict__ info, const int store_l,
                                                      double* __restrict__ ytop) {
  __shared__ ReducedLds<NX, NU, false> lds[4];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), b = blockIdx.x;
  // (more than four separators on a level -- a launch that starts below the last three levels --: in turn; one flat
  //  loop over (level, turn) so that the body is instantiated once)
  for (int l = l0, s0 = 0; l < d.K;) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));  // (opaque per round: keeps the lane-dependent addresses of the body out of the loop's preheader)
    const int cnt = d.N >> (l + 1);
    if (wave + s0 < cnt)
      reduced_separator_mc<NX, NU, true>(d, l, (wave + s0) * (2 << l), b, lane, AB, QR, rhs, red, rec, F, info, store_l,
                                         lds[wave]);
    s0 += 4;
    if (s0 < cnt) continue;  // (uniform: the same for every wavefront)
    s0 = 0;
    ++l;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's pushes and record are acknowledged
    __syncthreads();
  }
  // (the records of this launch were stored by wavefronts of this workgroup, acknowledged and behind a barrier; nobody
  //  on this CU has read those lines before: the plain loads of the sweep fetch them from L2)
  if (ytop) backsub_top_body<NX>(d, b, threadIdx.x, rec, ytop, reinterpret_cast<double*>(&lds[0]), -1, nullptr);
}

//   grid (N / 4, batch), block 64; N >= 8; instances with matrix-core products only.
// TREE: the wavefront does not stop after its level-1 separator. Every separator of level >= 2 has
// an arrival counter; a wavefront that has finished a subtree bumps the counter of the parent
// separator, and the one that arrives second -- both child subtrees are then eliminated and their
// pushes visible -- goes on to eliminate the parent (reduced_separator_mc), and so on up to the
// root. Nobody ever waits: the first arriver simply exits. One launch for the whole factorisation
// instead of 1 + (K - 2); the thinly populated upper levels of one problem overlap with the bottom
// levels of the next ones. The wavefront that eliminates the root resets its problem's counters (a launch
// that did not run to completion makes the host zero them before the next solve).
// Correct on every placement (write-through pushes, L1-bypassing slot loads, slots padded to whole
// lines; no fences: an agent-scope fence writes back / invalidates a whole L2 and made this 18 ms),
// Used for small batches (all bottom wavefronts resident at once: (6,3,256)x1 0.053 -> 0.046 ms);
// for large ones one launch per level is faster (1.07 vs 0.61 ms at (12,4,256)x1024: a climbing
// wavefront waits for its store acknowledgements, the counter round trip and the slot coming from
// memory, ~15 us per separator while it holds its SIMD slot). launch_small.hpp picks by batch size,
// NDLQR_TREE=0/1 overrides.
// Every accumulator element still receives its additions in a fixed order: its contributors are
// the separators along one spine of the subtree below it, and each of them finishes its pushes
// before its parent starts.
// The four knots k0 .. k0 + 3 of problem b by one wavefront: leaf phase + tree levels 0 and 1 (body of bottom_reduced_mc
// and of bottom8_reduced_mc). slotA / slotB: where the group's pushes to the separators k0 - 1 / k0 + 3 go instead of
// their slots in `red` (bottom8_reduced_mc: the level-2 separator between its two groups lives in LDS); null: `red`.
template <int NX, int NU, bool TREE, bool REDIRECT>
__device__ __forceinline__ void bottom_group_mc(const Dims& d, const int k0, const int b, const int lane,
                                                const double* __restrict__ AB, const double* __restrict__ QR,
                                                const double* __restrict__ rhs, double* red, double* __restrict__ rec,
                                                double* F, int* __restrict__ info, const int store_l, const int compact0,
                                                ReducedLds<NX, NU>& lds, double* slotA, double* slotB,
                                                const bool with_m, acc4_t* c_m) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, KSN = (NX + 3) / 4;
  constexpr int REC = 2 * NN + NX;
  // row pitch of the staged [A | B]: even W padded by two doubles so that the 16 rows an operand
  // fragment touches fall into distinct LDS banks
  constexpr int WP = ReducedLds<NX, NU>::WP;
  // the staged [A | B] is dead once the leaf tiles and coupling fragments are in registers: the
  // core's scratch lies over it
  double* abs_ = lds.buf;            // [A | B] of the four knots
  double* rq = lds.rq;               // 1 / [Q | R] of the four knots
  double* rh = lds.rh;               // their raw right-hand sides
  const int N = d.N;
  const int li = lane & 15, lk = lane >> 4;
  const int ri = li < NX ? li : NX - 1;
  const bool hasA = k0 > 0, hasB = k0 + 4 < N;  // separators k0 - 1 / k0 + 3 exist
  SEG_INIT();

  // ---- the wavefront's whole input (6.9 KB at (12, 4)) in ONE round of coalesced loads -- every
  //      load is issued before the first one is waited for --, staged in LDS; the operand fragments
  //      of the matrix-core products are gathered from there
  {
    const double* abm = AB + ((size_t)b * N + k0) * NX * W;  // four knots, contiguous, 16-byte aligned
    const double* q0 = QR + ((size_t)b * N + k0) * W;
    const double* r0 = rhs + ((size_t)b * N + k0) * ROWS;
    constexpr int NQ = 4 * W, IQ = (NQ + 63) / 64, NR = 4 * ROWS, IR = (NR + 63) / 64;
    double qv[IQ], rv[IR];
#pragma unroll
    for (int it = 0; it < IQ; ++it) { const int e = lane + 64 * it; qv[it] = q0[e < NQ ? e : NQ - 1]; }
#pragma unroll
    for (int it = 0; it < IR; ++it) { const int e = lane + 64 * it; rv[it] = r0[e < NR ? e : NR - 1]; }
    if constexpr (W % 2 == 0) {
      constexpr int NA = 4 * NX * W / 2, IA = (NA + 63) / 64;
      double2 t[IA];
#pragma unroll
      for (int it = 0; it < IA; ++it) {
        const int e = lane + 64 * it;
        t[it] = reinterpret_cast<const double2*>(abm)[e < NA ? e : NA - 1];
      }
#pragma unroll
      for (int it = 0; it < IA; ++it) {  // unconditional stores on the clamped index: see reduced_separator_mc
        const int e = lane + 64 * it, ec = e < NA ? e : NA - 1, row = ec / (W / 2), c2 = ec - row * (W / 2);
        reinterpret_cast<double2*>(&abs_[row * WP])[c2] = t[it];
      }
    } else {
      constexpr int NA = 4 * NX * W, IA = (NA + 63) / 64;
      double t[IA];
#pragma unroll
      for (int it = 0; it < IA; ++it) { const int e = lane + 64 * it; t[it] = abm[e < NA ? e : NA - 1]; }
#pragma unroll
      for (int it = 0; it < IA; ++it) { const int e = lane + 64 * it; abs_[e < NA ? e : NA - 1] = t[it]; }
    }
#pragma unroll
    for (int it = 0; it < IQ; ++it) {
      const int e = lane + 64 * it, ec = e < NQ ? e : NQ - 1, kn = ec / W, c = ec - kn * W;
      rq[ec] = 1.0 / qv[it];
      if (e < NQ && !(qv[it] > 0.0) && !(k0 + kn == N - 1 && c >= NX)) flag_failure(info, d, b);  // terminal R is unused
    }
#pragma unroll
    for (int it = 0; it < IR; ++it) { const int e = lane + 64 * it; rh[e < NR ? e : NR - 1] = rv[it]; }
  }
  wave_lds_sync();
  SEG(20);

  auto none = [](int) { return 0.0; };
  // (knot 0 -- fixed state -- is one wavefront in N / 4: a uniform branch keeps its selects out of everybody else's tile)
  acc4_t c_s0;
  if (k0 == 0) c_s0 = leaf_tile_mc<NX, NU, WP>(lane, true, abs_, rq, rq + W, rh, rh + ROWS, none);
  else c_s0 = leaf_tile_mc<NX, NU, WP>(lane, false, abs_, rq, rq + W, rh, rh + ROWS, none);
  acc4_t c_t = leaf_tile_mc<NX, NU, WP>(lane, false, abs_ + NX * WP, rq + W, rq + 2 * W, rh + ROWS, rh + 2 * ROWS, none);
  acc4_t c_s2 = leaf_tile_mc<NX, NU, WP>(lane, false, abs_ + 2 * NX * WP, rq + 2 * W, rq + 3 * W, rh + 2 * ROWS,
                                         rh + 3 * ROWS, none);
  if constexpr (REDIRECT) {
    // bottom8_reduced_mc, the wavefront that will eliminate the level-2 separator m = k0 + 3 behind the workgroup barrier:
    // the leaf tile of m now, while [A_m | B_m] is staged (no load from memory stands between the barrier and the
    // elimination: the other wavefront's slot idles meanwhile). Knot m + 1 is the other wavefront's: its weights and
    // right-hand side come from memory, one element per lane, and reach the lanes that need them by shuffles.
    if (with_m) {
      const double* qn = QR + ((size_t)b * N + k0 + 4) * W;      // Q_{m+1}
      const double* rn = rhs + ((size_t)b * N + k0 + 4) * ROWS;  // rhs(m+1): lambda | x | ..
      const int e = lane < 3 * NX ? lane : 3 * NX - 1;
      const double raw = e < NX ? qn[e] : rn[e - NX];
      const double val = e < NX ? 1.0 / raw : raw;  // (one division per wavefront, like the staged weights)
      double nw[4], nl[4], nxv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = lk + 4 * g, ic = i < NX ? i : NX - 1;
        nw[g] = __shfl(val, ic, 64); nl[g] = __shfl(val, NX + ic, 64); nxv[g] = __shfl(val, 2 * NX + ic, 64);
      }
      *c_m = leaf_tile_mc<NX, NU, WP, false>(lane, false, abs_ + 3 * NX * WP, rq + 3 * W, rq, rh + 3 * ROWS, rh, none);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = lk + 4 * g;
        (*c_m)[g] += li == NX ? -fma(nxv[g], nw[g], nl[g]) : (i == li ? nw[g] : 0.0);
      }
    }
  }
  acc4_t X0, X1, unused;
  double* myrec = rec + ((size_t)b * N + k0) * REC;
  // what the two level-0 separators hand to t and to the neighbours of the group (the Gram products take (R, X) or,
  // for compact records, (Y, Y))
  acc4_t park_a, ca_t, park_b11, cb_t;
  auto hook_s0 = [&](const double (&R0)[KSN], const double (&R1)[KSN], const acc4_t& Z0, const acc4_t& Z1) {
    acc4_t g11;  // s0 (left child of t): DL[t], gL[t], CA[t]; its a-side faces separator k0 - 1
    gram_mc<NX, true, false, true, true>(R0, R1, Z0, Z1, park_a, unused, ca_t, g11);
#pragma unroll
    for (int g = 0; g < 4; ++g) c_t[g] -= (li < NX) ? g11[g] : ca_t[g];  // column NX: Y_bb' y_z
  };
  auto hook_s2 = [&](const double (&R0)[KSN], const double (&R1)[KSN], const acc4_t& Z0, const acc4_t& Z1) {
    acc4_t g00;  // s2 (right child of t): DR[t], gR[t], CB[t]; bb-side faces separator k0 + 3
    gram_mc<NX, true, true, false, true>(R0, R1, Z0, Z1, g00, cb_t, unused, park_b11);
#pragma unroll
    for (int g = 0; g < 4; ++g) c_t[g] -= g00[g];  // Y_a' [Y_a | y_z]
  };

  // (the tree schedule never asks for compact records: its instantiation does not carry that path)
  const bool compact_path = !TREE && compact0 != 0;
  if (!TREE && compact_path) {
    // ---- compact records: the panel columns ride through the Cholesky pass, one per lane (chol_pair_y_mc)
    double wcol[NX];
    {
      const int h = lane & 31, x = lane >> 5;  // x = 0: s0 (knots k0, k0 + 1), 1: s2 (knots k0 + 2, k0 + 3)
      const bool is_a = h < NX, is_b = h > NX && h <= 2 * NX;
      const int ca = is_a ? h : 0, cb = is_b ? h - NX - 1 : 0;
      // r_a(k, ca) = -A_s(k, ca) / Q_s(ca): column ca of [A_s | B_s], stride WP, one weight;
      // r_bb(k, cb) = -A_{s+1}(cb, k) / Q_{s+1}(k): row cb of [A_{s+1} | B_{s+1}], stride 1, weight k
      const double* src = abs_ + (2 * x) * NX * WP + (is_a ? ca : NX * WP + cb * WP);
      const double* scl = rq + (2 * x) * W + (is_a ? ca : W);
      const int sstr = is_a ? WP : 1, wstr = is_a ? 0 : 1;
      const bool on = (is_a && (x == 1 || hasA)) || (is_b && (x == 0 || hasB));
      const double sign = on ? -1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) wcol[k] = (src[k * sstr] * scl[k * wstr]) * sign;
    }
    wave_lds_sync();  // last read of the staged [A | B]
    SEG(21);
    if (chol_pair_y_mc<NX>(lane, c_s0, c_s2, lds.buf, wcol, myrec, myrec + 2 * REC) && (lane == 0 || lane == 32))
      flag_failure(info, d, b);
    SEG(30);
    using PY = McPairYLayout<NX>;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const double* Y0 = lds.buf + PY::tile(x, 0);
      const double* Y1 = lds.buf + PY::tile(x, 1);
      double y0[KSN], y1[KSN];
      acc4_t Z0 = {0.0, 0.0, 0.0, 0.0}, Z1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < KSN; ++q) {
        y0[q] = Y0[(4 * q + lk) * PY::YP + li];
        y1[q] = Y1[(4 * q + lk) * PY::YP + li];
        Z0[q] = y0[q]; Z1[q] = y1[q];
      }
      if (x == 0) hook_s0(y0, y1, Z0, Z1); else hook_s2(y0, y1, Z0, Z1);
    }
    SEG(35);
  } else {
  // operand fragments of the couplings: r_a(i, j) = -A_s(i, j) / Q_s(j), r_bb(i, j) = -A_{s+1}(j, i) / Q_{s+1}(i)
  double ra0[KSN], rb0[KSN], ra2[KSN], rb2[KSN];
#pragma unroll
  for (int q = 0; q < KSN; ++q) {
    const int kq = 4 * q + lk, i = kq < NX ? kq : NX - 1;
    const double a0 = abs_[i * WP + ri], b0 = abs_[(NX + ri) * WP + i];
    const double a2 = abs_[(2 * NX + i) * WP + ri], b2 = abs_[(3 * NX + ri) * WP + i];
    const double s0 = rq[ri], s1 = rq[W + i], s2 = rq[2 * W + ri], s3 = rq[3 * W + i];
    ra0[q] = hasA ? -a0 * s0 : 0.0;
    rb0[q] = -b0 * s1;
    ra2[q] = -a2 * s2;
    rb2[q] = hasB ? -b2 * s3 : 0.0;
  }
  wave_lds_sync();  // last read of the staged [A | B]
  SEG(21);

  // ---- Cholesky + inverse of the two level-0 separators s0 = k0 and s2 = k0 + 2 in ONE pass (DPP rows 0-1 / 2-3)
  using Pair = McPairLayout<NX>;
  if (chol_pair_mc<NX>(lane, c_s0, c_s2, lds.buf, store_l ? Fblk(F, d, b, 0, k0 + 1) : nullptr,
                       store_l ? Fblk(F, d, b, 0, k0 + 3) : nullptr) &&
      (lane == 0 || lane == 32))
    flag_failure(info, d, b);
  SEG(30);
  factor_tail_mc<NX>(lane, c_s0, ra0, rb0, lds.buf + Pair::W_A, X0, X1, hook_s0);
  store_record_mc<NX>(myrec, lane, hasA, true, X0, X1);
  factor_tail_mc<NX>(lane, c_s2, ra2, rb2, lds.buf + Pair::W_B, X0, X1, hook_s2);
  store_record_mc<NX>(myrec + 2 * REC, lane, true, hasB, X0, X1);
  SEG(35);
  }
  // ---- t = k0 + 1 (level 1): r_a = -CA[t] = -Y_bb'Y_a of s0, r_bb = -CB[t] = -Y_a'Y_bb of s2: the two coupling tiles
  //      go from the accumulators through LDS (over the dead tiles of the pair) to one panel column per lane
  //      (chol_wy_mc); pushes of the whole group to the separators k0 - 1 (A) and k0 + 3 (B)
  double wcol[NX];
  {
    constexpr int TP = 17;
    static_assert(2 * 16 * TP <= ReducedLds<NX, NU>::NBUF, "both coupling tiles fit the buffer");
    double* ta = lds.buf;
    double* tb = lds.buf + 16 * TP;
    wave_lds_sync();  // (the operands of the level-0 tails are in registers)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ta[(lk + 4 * g) * TP + li] = -ca_t[g];
      tb[(lk + 4 * g) * TP + li] = -cb_t[g];
    }
    wave_lds_sync();
    const int h = lane & 31;
    const bool is_a = h < NX, is_b = h > NX && h <= 2 * NX;
    const double* src = is_a ? ta + h : tb + (is_b ? h - NX - 1 : 0);  // (the lanes without a column: finite don't-cares)
#pragma unroll
    for (int k = 0; k < NX; ++k) wcol[k] = src[k * TP];
    wave_lds_sync();  // the S-bar tile of the pass goes over the coupling tiles
  }
  const bool leftchild = (k0 & 4) == 0;
  RedSlot<NX> sa = red_slot<NX>(red, d, b, hasA ? k0 - 1 : 3);
  RedSlot<NX> sb = red_slot<NX>(red, d, b, hasB ? k0 + 3 : 3);
  if constexpr (REDIRECT) {
    if (slotA) sa.p = slotA;
    if (slotB) sb.p = slotB;
  }
  if (chol_wy_mc<NX>(lane, c_t, lds.buf, wcol, lstore_of<NX>(store_l, F, rec, d, b, 1, k0 + 1)) && lane == 0)
    flag_failure(info, d, b);
  tail_wy_mc<NX>(lane, lds.buf, X0, X1,
                 [&](const double (&R0)[KSN], const double (&R1)[KSN], const acc4_t& Z0, const acc4_t& Z1) {
                   acc4_t g00, g01, g11;
                   gram_mc<NX, true, true, false, true>(R0, R1, Z0, Z1, g00, g01, unused, g11);
                   if constexpr (TREE)
                     push_mc<NX>(lane, hasA, hasB, leftchild, sa, sb, g00, g01, g11, park_a, cb_t, park_b11,
                                 StoreThrough(), StoreThrough());
                   else
                     push_mc<NX>(lane, hasA, hasB, leftchild, sa, sb, g00, g01, g11, park_a, cb_t, park_b11,
                                 StorePlain(), StorePlain());
                 });
  SEG(34);
  store_record_mc<NX>(myrec + REC, lane, hasA, hasB, X0, X1);
#ifdef NDLQR_SEGTIME
  __builtin_amdgcn_s_waitcnt(0);
#endif
  SEG(36);
}

// COMPACT: the instantiation of the default schedule (compact level-0 records: `compact0` is taken as 1 and the
// full-record path is not compiled in -- 104 instead of 126 VGPRs at (12,4), no spills at (13,4) and (15,2)).
template <int NX, int NU, bool TREE, bool COMPACT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void bottom_reduced_mc(
    Dims d, const double* __restrict__ AB, const double* __restrict__ QR, const double* __restrict__ rhs,
    double* red, double* __restrict__ rec, double* F, int* __restrict__ info, const int store_l, int* cnt,
    const int compact0) {
  static_assert(!(TREE && COMPACT), "the tree schedule keeps full records");
  __shared__ ReducedLds<NX, NU> lds;
  const int lane = threadIdx.x, b = blockIdx.y, N = d.N, k0 = (blockIdx.x + d.xoff) * 4;
  bottom_group_mc<NX, NU, TREE, false>(d, k0, b, lane, AB, QR, rhs, red, rec, F, info, store_l, COMPACT ? 1 : compact0, lds,
                                       nullptr, nullptr);

  if constexpr (TREE) {
    int l = 1, base = k0;  // finished: the level-l separator of subtree [base, base + 2^(l+1))
    for (;;) {
      const int T = 2 << l;
      if (T >= N) {  // that was the root: every arrival of this problem has been counted -- leave the counters at zero
        for (int e = lane; e < (N >> 2); e += 64) cnt[(size_t)b * (N >> 2) + e] = 0;
        break;
      }
      const bool left = (base & T) == 0;
      const int p = left ? base + T - 1 : base - 1;  // the parent separator (level l + 1)
      // every push of this wavefront has left the chip-side caches (write-through stores, atomics)
      // and is acknowledged before the arrival is counted
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int old = 0;
      if (lane == 0) old = atomicAdd(cnt + (size_t)b * (N >> 2) + (p >> 2), 1);
      old = __builtin_amdgcn_readfirstlane(old);
      if ((old & 1) == 0) break;  // the sibling subtree is still being eliminated: its wavefront goes on
      asm volatile("" ::: "memory");
      l = l + 1;
      base = left ? base : base - T;
      wave_lds_sync();
      // (the lane id is made opaque per round: otherwise every lane-dependent address of the body is hoisted out of
      //  the loop -- 45 VGPRs of them ended up in scratch memory)
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      reduced_separator_mc<NX, NU, true>(d, l, base, b, lane_s, AB, QR, rhs, red, rec, F, info, store_l, lds);
    }
  }
}

}  // namespace ndlqr
