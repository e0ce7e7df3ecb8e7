// kernels_rowbcast.hpp -- fast mode without KEEP, separator-only (block cyclic reduction) schedule,
// third form of the separator core: ONE SEPARATOR PER 16-LANE DPP ROW, four per wavefront.
//
// Same mathematics as kernels_bottom_reduced.hpp (see its header and DESIGN.md section 2; reference:
// ndlqr_SolveLeaf src/nested_dissection.c:10-105, ndlqr_FactorInnerProduct :114-134, the separator
// Cholesky of src/solve.c:87-98, ndlqr_SolveCholeskyFactor :136-152, ndlqr_UpdateShurFactor :154-171).
// What changes is where the numbers live. Lane i of a row holds ROW i of every matrix of "its"
// separator (a matrix with C columns = C registers), and every product is of the form
//     out(i, c) += A(i, k) * B(k, c)   =   v_fmac_f64_dpp out[c], B[c], A[k]  row_newbcast:k
// -- the broadcast of row k of B from lane k is part of the FMA (DPP source modifier), so there is
// no v_readlane, no LDS panel, no matrix-core tile padding and no idle replication: measured on
// gfx950 (tools/ubench/issue_rates.hip, profiles/r02_issue_rates.txt) a v_fmac_f64_dpp costs exactly
// one plain v_fma_f64 issue slot, a 2 x v_readlane + v_fma_f64 broadcast 3.3 slots, and the fp64
// matrix cores have no rate advantage (16x16x4: 14.6 slots for 2048 flops; 4x4x4: 3.7 for 512) while
// 12-wide blocks fill only 42-58 % of a 16x16 tile. Per separator the row form needs ~1 300
// instructions for FOUR separators; the matrix-core form needed 1 715 vector + 66 x 14.6 matrix slots
// for THREE.
//
//   rb_bottom       leaf phase + tree levels 0, 1: a wavefront owns 16 knots = four groups of four, one
//                   group per DPP row; three passes (s0 = k0, s2 = k0 + 2, t = k0 + 1 of every group)
//   rb_backsub_top  multipliers of the separators of level >= 3, one workgroup per problem
//   rb_backsub      back-substitution of eight knots per workgroup; level-0 separators keep only the
//                   Cholesky factor of S-bar (packed lower triangle, n (n + 1) / 2 doubles instead of
//                   2 n^2 + n): their f_a, f_bb, z_sep follow from the problem data the kernel reads anyway
// The upper levels stay on reduced_level_mc (kernels_bottom_reduced.hpp): a four-separators-per-
// wavefront level kernel in this form was measured slower at every level (127 vs 108 us at level 2,
// 13 vs 6.5 us at the root: 32 KB of staged slots per wavefront leave five wavefronts per CU).
//
// Transposed operands (r_a', r_bb') never need a transposition: at level 0 they are other views of
// the staged [A | B] (r_bb(s)' = r_a(s + 1) by symmetry of the reduced system), at level 1 both
// orientations of the coupling blocks are computed (G10 and G01 = G10'), above that the slot could be
// read by rows or by columns.
#pragma once
#include "kernels_dpp.hpp"
#include "kernels_small.hpp"

namespace ndlqr {

// out[c] (+)= (-) sum_{k < KD} A[k] * B(k, c), c < NC: A = this lane's row of the left factor, B[c] = this
// lane's row of the right factor (row k is taken from lane k). INIT: out starts from zero; NEG: the
// product is subtracted. Four columns per asm statement: fewer statements for the compiler's hazard
// recognizer to pad.
template <int K, bool NEG>
__device__ __forceinline__ void fmac_bc4(double& a0, double& a1, double& a2, double& a3, const double x0, const double x1,
                                         const double x2, const double x3, const double y) {
  if constexpr (NEG)
    asm volatile("v_fmac_f64_dpp %0, -%4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, -%5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, -%6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, -%7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y), "n"(K));
  else
    asm volatile("v_fmac_f64_dpp %0, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y), "n"(K));
}

template <int KD, int NC, bool INIT, bool NEG = false, int NA, int NB>
__device__ __forceinline__ void rb_mul(const double (&A)[NA], double (&B)[NB], double (&out)[NC]) {
  static_assert(KD <= NA && NC <= NB && KD <= 16, "operand shapes");
  if constexpr (INIT) {
#pragma unroll
    for (int c = 0; c < NC; ++c) out[c] = 0.0;
  }
  dpp_fence(B);
  sfor<KD>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
#pragma unroll
    for (int c = 0; c + 4 <= NC; c += 4) fmac_bc4<k, NEG>(out[c], out[c + 1], out[c + 2], out[c + 3], B[c], B[c + 1], B[c + 2], B[c + 3], A[k]);
#pragma unroll
    for (int c = NC / 4 * 4; c < NC; ++c) {
      if constexpr (NEG) fnmac_bc<k>(out[c], B[c], A[k]); else fmac_bc<k>(out[c], B[c], A[k]);
    }
  });
}
// the same for one column
template <int KD, int NA>
__device__ __forceinline__ double rb_mulv(const double (&A)[NA], double b, double out = 0.0) {
  dpp_fence(b);
  sfor<KD>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    fmac_bc<k>(out, b, A[k]);
  });
  return out;
}
// out - sum_k A[k] * b(k)
template <int KD, int NA>
__device__ __forceinline__ double rb_mulv_sub(const double (&A)[NA], double b, double out) {
  dpp_fence(b);
  sfor<KD>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    fnmac_bc<k>(out, b, A[k]);
  });
  return out;
}

// ------------------------------------------------------------------------------------- separator core (rb_chol_inv: kernels_dpp.hpp)
// X = S-bar^-1 R without ever forming S-bar^-1 (round 4; DESIGN.md section 3.2 "numerics"): after rb_chol_inv lane i
// holds row i of L and column i of W = L^-1.
//   rb_solve_prep   dk = 1 / L(k, k) (= W(k, k), from lane k), m[k] = L(i, k) dk[k] for k < i (else 0),
//                   wt[k] = W(k, i) dk[k]
//   rb_fsub         Z <- D L^-1 Z in place, rows in lanes (right-looking forward substitution: step k subtracts
//                   m[k] times row k -- final by then, taken from lane k inside the FMA -- from every row below it;
//                   the scaling by 1 / L(k, k) is folded into m and wt, so no step rescales anything)
//   X = rb_mul(wt, Z)   = W' D^-1 (D L^-1 R) = L^-T L^-1 R
// n (2n + 1) more row-broadcast FMAs per separator than X = (W'W) R, minus the n (n + 1) / 2 of W'W.
template <int NX>
__device__ __forceinline__ void rb_solve_prep(const int i, const double (&Lrow)[NX], double (&w)[NX], double (&m)[NX],
                                              double (&wt)[NX]) {
  dpp_fence(w);
  sfor<NX>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double dk = row_bc<k>(w[k]);
    m[k] = (k < i) ? Lrow[k] * dk : 0.0;
    wt[k] = w[k] * dk;
  });
}
template <int KD, int NC, int NA, int NB>
__device__ __forceinline__ void rb_fsub(const double (&m)[NA], double (&Z)[NB]) {
  static_assert(KD <= NA && NC <= NB && NC >= 4, "shapes; a column is read by a DPP instruction at least NC instructions after it was written");
  dpp_fence(Z);
  sfor<KD>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
#pragma unroll
    for (int c = 0; c < NC; ++c) fnmac_bc<k>(Z[c], Z[c], m[k]);
  });
}

// [S-bar | b~] of a separator from the staged problem data (lane i: row i; see leaf_tile_mc):
//   abrow  row i of [A_s | B_s]                 T[c] = [A_s | B_s](c, i) / [Q_s | R_s](i)  (lane i < W)
//   tz     z-hat(i): rhs column operand         q1 = 1 / Q_{s+1}(i),  z1l, z1x: rhs(s+1).lambda / .x (i)
template <int NX, int W>
__device__ __forceinline__ void rb_leaf(const int i, const double (&abrow)[W], double (&T)[NX], const double tz,
                                        const double q1, const double z1l, const double z1x, double (&S)[NX],
                                        double& bz) {
  rb_mul<W, NX, true>(abrow, T, S);
  bz = rb_mulv<W>(abrow, tz) - fma(z1x, q1, z1l);
#pragma unroll
  for (int c = 0; c < NX; ++c) S[c] += (c == i) ? q1 : 0.0;
}

// LDS image of one staged knot: [A | B] with a row pitch that keeps 16-byte row reads of 12..16
// consecutive rows on distinct banks, reciprocal weights, raw right-hand side.
template <int NX, int NU>
struct RbKnot {
  static constexpr int W = NX + NU, ROWS = 2 * NX + NU, WP = (W % 2 == 0) ? W + 2 : W + 1;
};

// [A | B] of knots 4 g + 2 ph, 4 g + 2 ph + 1 (g = 0..3) of the wavefront's sixteen, coalesced 16-byte
// loads -> LDS slots [2 g + j]; every load is issued before the first store.
template <int NX, int NU>
__device__ __forceinline__ void rb_stage_ab(const int lane, const int ph, const double* __restrict__ abm,
                                            double (&ab)[8][NX * RbKnot<NX, NU>::WP]) {
  constexpr int W = NX + NU, WP = RbKnot<NX, NU>::WP;
  if constexpr (W % 2 == 0) {
    constexpr int PK = NX * W / 2, NA = 8 * PK, IA = (NA + 63) / 64;  // double2 words per knot / in all
    double2 t[IA];
#pragma unroll
    for (int it = 0; it < IA; ++it) {
      const int e = lane + 64 * it, ec = e < NA ? e : NA - 1;
      const int slot = ec / PK, within = ec - slot * PK;
      const int knot = 4 * (slot >> 1) + 2 * ph + (slot & 1);
      t[it] = reinterpret_cast<const double2*>(abm)[knot * PK + within];
    }
#pragma unroll
    for (int it = 0; it < IA; ++it) {  // unconditional stores on the clamped index (see reduced_separator_mc)
      const int e = lane + 64 * it, ec = e < NA ? e : NA - 1;
      const int slot = ec / PK, within = ec - slot * PK, row = within / (W / 2), c2 = within - row * (W / 2);
      reinterpret_cast<double2*>(&ab[slot][row * WP])[c2] = t[it];
    }
  } else {
    constexpr int PK = NX * W, NA = 8 * PK, IA = (NA + 63) / 64;
    double t[IA];
#pragma unroll
    for (int it = 0; it < IA; ++it) {
      const int e = lane + 64 * it, ec = e < NA ? e : NA - 1;
      const int slot = ec / PK, within = ec - slot * PK;
      const int knot = 4 * (slot >> 1) + 2 * ph + (slot & 1);
      t[it] = abm[knot * PK + within];
    }
#pragma unroll
    for (int it = 0; it < IA; ++it) {
      const int e = lane + 64 * it, ec = e < NA ? e : NA - 1;
      const int slot = ec / PK, within = ec - slot * PK, row = within / W, c = within - row * W;
      ab[slot][row * WP + c] = t[it];
    }
  }
}

// ------------------------------------------------------------------------------------- bottom levels
//   grid (N / 16, batch), block 64; N >= 16; NX <= 16, NX + NU <= 16.
// Row g of the wavefront owns knots k0 .. k0 + 3, k0 = 16 blockIdx.x + 4 g. Pass 1: s0 = k0 of every
// group from knots k0, k0 + 1 (and the leaf tile of t = k0 + 1, whose [A | B] is staged then);
// pass 2: s2 = k0 + 2 from knots k0 + 2, k0 + 3 (re-staged over the first two); pass 3: t. What the
// level-0 separators contribute to t stays in registers; what the group contributes to the
// separators k0 - 1 and k0 + 3 goes to their slots: the level-0 share as plain stores (which
// initialise the accumulators of this solve), t's share as atomic adds behind them.
template <int NX, int NU>
struct alignas(16) RbBottomLds {
  static constexpr int W = NX + NU, ROWS = 2 * NX + NU, WP = RbKnot<NX, NU>::WP;
  double ab[8][NX * WP];   // two knots of each of the four groups: [2 g + j]
  double rq[16][W];        // 1 / [Q | R] of all sixteen knots
  double rh[16][ROWS];     // raw right-hand sides of all sixteen knots
};

template <int NX, int NU>
__global__ __launch_bounds__(64, 2) void rb_bottom(Dims d, const double* __restrict__ AB, const double* __restrict__ QR,
                                                   const double* __restrict__ rhs, double* red,
                                                   double* __restrict__ rec, int* __restrict__ info) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, REC = 2 * NN + NX, WP = RbKnot<NX, NU>::WP;
  static_assert(NX <= 16 && W <= 16, "one DPP row holds the rows of S-bar and of [A | B]'");
  __shared__ RbBottomLds<NX, NU> lds;
  const int lane = threadIdx.x, b = blockIdx.y, N = d.N;
  const int g = lane >> 4, i = lane & 15;
  const int ic = i < NX ? i : NX - 1;  // rows >= NX of a DPP row are padding: they repeat row NX - 1
  const int kc = i < W ? i : W - 1;
  const int kw = (blockIdx.x + d.xoff) * 16;  // first knot of the wavefront
  const int k0 = kw + 4 * g;           // first knot of this row's group
  const bool hasA = k0 > 0, hasB = k0 + 4 < N, first = k0 == 0;
  const bool rowlane = i < NX;

  // ---- staging: weights and right-hand sides of all 16 knots, [A | B] of knots k0 + 2 ph, k0 + 2 ph + 1
  const double* const abm = AB + ((size_t)b * N + kw) * NX * W;
  {
    const double* q0 = QR + ((size_t)b * N + kw) * W;
    const double* r0 = rhs + ((size_t)b * N + kw) * ROWS;
    constexpr int NQ = 16 * W, IQ = (NQ + 63) / 64, NR = 16 * ROWS, IR = (NR + 63) / 64;
    double qv[IQ], rv[IR];
#pragma unroll
    for (int it = 0; it < IQ; ++it) { const int e = lane + 64 * it; qv[it] = q0[e < NQ ? e : NQ - 1]; }
#pragma unroll
    for (int it = 0; it < IR; ++it) { const int e = lane + 64 * it; rv[it] = r0[e < NR ? e : NR - 1]; }
    rb_stage_ab<NX, NU>(lane, 0, abm, lds.ab);
#pragma unroll
    for (int it = 0; it < IQ; ++it) {
      const int e = lane + 64 * it, ec = e < NQ ? e : NQ - 1, kn = ec / W, c = ec - kn * W;
      (&lds.rq[0][0])[ec] = 1.0 / qv[it];
      if (e < NQ && !(qv[it] > 0.0) && !(kw + kn == N - 1 && c >= NX)) flag_failure(info, d, b);  // terminal R is unused
    }
#pragma unroll
    for (int it = 0; it < IR; ++it) { const int e = lane + 64 * it; (&lds.rh[0][0])[e < NR ? e : NR - 1] = rv[it]; }
  }
  wave_lds_sync();

  // operands of the separator between staged knots `sa` (knot index kn within the 16) and `sb`:
  //   abrow, T, tz for the leaf tile; Ra / RaT rows of r_a, r_a'; Rb / RbT rows of r_bb, r_bb'
  auto leaf_ops = [&](const int slot, const int kn, const bool fst, double (&abrow)[W], double (&T)[NX], double& tz) {
    const double* am = lds.ab[slot];
    if constexpr (WP % 2 == 0 && W % 2 == 0) {
#pragma unroll
      for (int k = 0; k < W; k += 2) {
        const double2 t = *reinterpret_cast<const double2*>(&am[ic * WP + k]);
        abrow[k] = t.x; abrow[k + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) abrow[k] = am[ic * WP + k];
    }
    const bool fx = fst && kc < NX;  // knot 0: the state is fixed, its columns leave S-bar and carry x0 in the rhs
    const double wk = fx ? 0.0 : lds.rq[kn][kc];
#pragma unroll
    for (int c = 0; c < NX; ++c) T[c] = am[c * WP + kc] * wk;
    tz = fx ? -lds.rh[kn][kc] : lds.rh[kn][NX + kc] * wk;
  };

  // Rows of the coupling blocks, re-read from the staged [A | B] right where they are needed (the live
  // ranges decide whether the kernel fits two wavefronts per SIMD): knot s in slot `sl` (knot index kn
  // of the sixteen), knot s + 1 in slot sl + 1.
  //   r_a(i, c)  = -A_s(i, c) / Q_s(c)            r_a(c, i)  = -A_s(c, i) / Q_s(i)
  //   r_bb(i, c) = -A_{s+1}(c, i) / Q_{s+1}(i)    r_bb(c, i) = -A_{s+1}(i, c) / Q_{s+1}(c)
  auto load_Ra = [&](const int sl, const int kn, const bool has, double (&R)[NX]) {
    const double* am = lds.ab[sl];
#pragma unroll
    for (int c = 0; c < NX; ++c) R[c] = has ? -am[ic * WP + c] * lds.rq[kn][c] : 0.0;
  };
  auto load_RaT = [&](const int sl, const int kn, const bool has, double (&R)[NX]) {
    const double* am = lds.ab[sl];
    const double wi = lds.rq[kn][ic];
#pragma unroll
    for (int c = 0; c < NX; ++c) R[c] = has ? -am[c * WP + ic] * wi : 0.0;
  };
  auto load_Rb = [&](const int sl, const int kn, const bool has, double (&R)[NX]) {
    const double* a1 = lds.ab[sl + 1];
    const double wi = lds.rq[kn + 1][ic];
#pragma unroll
    for (int c = 0; c < NX; ++c) R[c] = has ? -a1[c * WP + ic] * wi : 0.0;
  };
  auto load_RbT = [&](const int sl, const int kn, const bool has, double (&R)[NX]) {
    const double* a1 = lds.ab[sl + 1];
#pragma unroll
    for (int c = 0; c < NX; ++c) R[c] = has ? -a1[ic * WP + c] * lds.rq[kn + 1][c] : 0.0;
  };

  double St[NX], bzt;                             // [S-bar | b~] of t, accumulated over passes 1 and 2
  double Rat[NX], RaTt[NX], Rbt[NX], RbTt[NX];    // rows of r_a, r_a', r_bb, r_bb' of t
  double* const myrec = rec + ((size_t)b * N + k0) * REC;
  const RedSlot<NX> sA = red_slot<NX>(red, d, b, hasA ? k0 - 1 : 3);
  const RedSlot<NX> sB = red_slot<NX>(red, d, b, hasB ? k0 + 3 : 3);

  // Elimination of one level-0 separator (knot s in slot sl): its Cholesky factor L goes to the record (row i, lower
  // triangle packed: entry (i, c), c <= i, at i (i + 1) / 2 + c; rb_backsub substitutes with it); returns
  // X = S-bar^-1 [r_a | b~ | r_bb] = L^-T L^-1 [..] (Xa[NX] = the rhs column).
  auto eliminate0 = [&](const int sl, const int kn, const bool fst, const bool ha, const bool hb, double* const r,
                        double (&Xa)[NX + 1], double (&Xb)[NX]) {
    double m[NX], wt[NX], bz;
    {
      double S[NX], w[NX];
      {
        double abrow[W], T[NX], tz;
        leaf_ops(sl, kn, fst, abrow, T, tz);
        rb_leaf<NX, W>(i, abrow, T, tz, lds.rq[kn + 1][ic], lds.rh[kn + 1][ic], lds.rh[kn + 1][NX + ic], S, bz);
      }
      if (rb_chol_inv<NX>(i, S, w) && i == 0) flag_failure(info, d, b);
      if (rowlane) {
#pragma unroll
        for (int c = 0; c < NX; ++c)
          if (c <= i) r[i * (i + 1) / 2 + c] = S[c];
      }
      rb_solve_prep<NX>(i, S, w, m, wt);
    }
    {
      double Z[NX + 1];
      {
        double R[NX];
        load_Ra(sl, kn, ha, R);
#pragma unroll
        for (int c = 0; c < NX; ++c) Z[c] = R[c];
      }
      Z[NX] = bz;
      rb_fsub<NX, NX + 1>(m, Z);
      rb_mul<NX, NX + 1, true>(wt, Z, Xa);
    }
    {
      double Z[NX];
      load_Rb(sl, kn, hb, Z);
      rb_fsub<NX, NX>(m, Z);
      rb_mul<NX, NX, true>(wt, Z, Xb);
    }
  };

  // ================================================================ pass 1: s0 = k0 (knots k0, k0 + 1)
  {
    // leaf tile of t first, while its [A | B] (knot k0 + 1) is staged
    double abrow[W], T[NX], tz;
    leaf_ops(2 * g + 1, 4 * g + 1, false, abrow, T, tz);
    rb_leaf<NX, W>(i, abrow, T, tz, lds.rq[4 * g + 2][ic], lds.rh[4 * g + 2][ic], lds.rh[4 * g + 2][NX + ic], St, bzt);
  }
  {
    double Xa[NX + 1], Xb[NX];
    eliminate0(2 * g, 4 * g, first, hasA, true, myrec, Xa, Xb);
    const double xz = Xa[NX];
    double Rt[NX], G[NX];
    load_RaT(2 * g, 4 * g, hasA, Rt);
    // to separator k0 - 1 (DR, gR: plain stores, they start this solve's accumulators) ...
    rb_mul<NX, NX, true>(Rt, Xa, G);                       // r_a' S^-1 r_a
    const double gv = rb_mulv<NX>(Rt, xz);                 // r_a' S^-1 b~
    if (hasA && rowlane) {
#pragma unroll
      for (int c = 0; c < NX; ++c)  // (row i of the symmetric block: its lower-triangle part, packed)
        if (c <= i) sA.DR()[i * (i + 1) / 2 + c] = G[c];
      sA.gR()[i] = gv;
    }
    rb_mul<NX, NX, true, true>(Rt, Xb, RaTt);              // -(r_a' S^-1 r_bb)(i, c) = r_a(t)(c, i)
    // ... and to t: DL[t] = r_bb' S^-1 r_bb, gL[t] = r_bb' S^-1 b~, r_a(t) = -r_bb' S^-1 r_a
    load_RbT(2 * g, 4 * g, true, Rt);
    rb_mul<NX, NX, false, true>(Rt, Xb, St);
    bzt = rb_mulv_sub<NX>(Rt, xz, bzt);
    rb_mul<NX, NX, true, true>(Rt, Xa, Rat);
  }

  // ================================================================ pass 2: s2 = k0 + 2 (knots k0 + 2, k0 + 3)
  wave_lds_sync();  // last read of the first two knots
  rb_stage_ab<NX, NU>(lane, 1, abm, lds.ab);
  wave_lds_sync();
  {
    double Xa[NX + 1], Xb[NX];
    eliminate0(2 * g, 4 * g + 2, false, true, hasB, myrec + 2 * REC, Xa, Xb);
    const double xz = Xa[NX];
    double Rt[NX], G[NX];
    load_RbT(2 * g, 4 * g + 2, hasB, Rt);
    rb_mul<NX, NX, true>(Rt, Xb, G);                       // r_bb' S^-1 r_bb -> DL of separator k0 + 3
    const double gv = rb_mulv<NX>(Rt, xz);
    if (hasB && rowlane) {
#pragma unroll
      for (int c = 0; c < NX; ++c)
        if (c <= i) sB.DL()[i * (i + 1) / 2 + c] = G[c];
      sB.gL()[i] = gv;
    }
    rb_mul<NX, NX, true, true>(Rt, Xa, RbTt);              // -(r_bb' S^-1 r_a)(i, c) = r_bb(t)(c, i)
    load_RaT(2 * g, 4 * g + 2, true, Rt);
    rb_mul<NX, NX, false, true>(Rt, Xa, St);               // DR[t], gR[t]
    bzt = rb_mulv_sub<NX>(Rt, xz, bzt);
    rb_mul<NX, NX, true, true>(Rt, Xb, Rbt);               // r_bb(t) = -r_a' S^-1 r_bb
  }

  // ================================================================ pass 3: t = k0 + 1
  {
    double Xa[NX + 1], Xb[NX];
    {
      double m[NX], wt[NX];
      {
        double w[NX];
        if (rb_chol_inv<NX>(i, St, w) && i == 0) flag_failure(info, d, b);
        rb_solve_prep<NX>(i, St, w, m, wt);
      }
      {
        double Z[NX + 1];
#pragma unroll
        for (int c = 0; c < NX; ++c) Z[c] = Rat[c];
        Z[NX] = bzt;
        rb_fsub<NX, NX + 1>(m, Z);
        rb_mul<NX, NX + 1, true>(wt, Z, Xa);
      }
      rb_fsub<NX, NX>(m, Rbt);
      rb_mul<NX, NX, true>(wt, Rbt, Xb);
    }
    const double xz = Xa[NX];
    if (rowlane) {  // record f_a | f_bb | z_sep
      double* r = myrec + REC;
      if (hasA) store_row<NX>(r + i * NX, reinterpret_cast<const double (&)[NX]>(Xa));
      if (hasB) store_row<NX>(r + NN + i * NX, Xb);
      r[2 * NN + i] = xz;
    }
    const bool leftchild = (k0 & 4) == 0;
    // DR / DL are symmetric (up to rounding) and kept as packed lower triangles: lane i adds its entries (c, i), c >= i,
    // into row c of the triangle, so that the lanes of a row hit one contiguous run (the memory-side atomic units take
    // one request per 64-byte segment)
    double G[NX];
    rb_mul<NX, NX, true>(RaTt, Xa, G);
    double gv = rb_mulv<NX>(RaTt, xz);
    if (hasA && rowlane) {
#pragma unroll
      for (int c = 0; c < NX; ++c)
        if (c >= i) atomicAdd(sA.DR() + c * (c + 1) / 2 + i, G[c]);
      atomicAdd(sA.gR() + i, gv);
    }
    rb_mul<NX, NX, true>(RaTt, Xb, G);  // coupling of the parent to its other neighbour: CA[B] = (r_a' S^-1 r_bb)' or CB[A] = r_a' S^-1 r_bb
    if (hasA && hasB && rowlane) {
      if (leftchild) {
#pragma unroll
        for (int c = 0; c < NX; ++c) sB.CA()[c * NX + i] = G[c];
      } else {
        store_row<NX>(sA.CB() + i * NX, G);
      }
    }
    rb_mul<NX, NX, true>(RbTt, Xb, G);
    gv = rb_mulv<NX>(RbTt, xz);
    if (hasB && rowlane) {
#pragma unroll
      for (int c = 0; c < NX; ++c)
        if (c >= i) atomicAdd(sB.DL() + c * (c + 1) / 2 + i, G[c]);
      atomicAdd(sB.gL() + i, gv);
    }
  }
}

// ------------------------------------------------------------------------------------- back-substitution
// Two launches. rb_backsub_top: multipliers of the separators of level >= 3, top-down over the tree,
//     y_s = z_sep(s) - f_a(s) y_A - f_bb(s) y_B      (A / B: the separators left / right of s's subtree),
// one workgroup per problem (N / 8 - 1 separators, K - 3 dependent steps), result in ytop[b][s >> 3].
// rb_backsub: one workgroup per eight knots = one level-2 subtree. It needs only the two multipliers
// next to it from ytop (backsub_small fetched the K - 3 records on its path to the root instead and
// resolved them again in every workgroup), resolves its level-2 and level-1 separators from their
// records and its four level-0 separators from the problem data and their compact record, the Cholesky factor L
// of S-bar (their neighbours s - 1, s + 1 are known by then):
//   v = [A_s | B_s] z-hat(s) - r_a y_{s-1} - r_bb y_{s+1} - z(s+1).lambda - z(s+1).x / Q_{s+1},   y_s = L^-T L^-1 v
//   r_a y = -A_s (y / Q_s),   r_bb y = -(A_{s+1}' y) / Q_{s+1}
// (A_{s+1}' y_{s+1} is the dot product the state rows of knot s + 1 need anyway); then states and
// inputs from the stationarity rows (src/solve.c:137-182 produces the same quantities level by level):
//   lambda_k = y_{k-1}; knot 0: lambda = Q x0 + q + A_0' y_0;
//   x_k = Q^-1 (-q_k - A_k' y_k + y_{k-1}),  u_k = R^-1 (-r_k - B_k' y_k).
//   rb_backsub_top: grid (batch), block 256, dynamic LDS (N / 8) * NX doubles; N >= 16.
// (device function: called by the kernel below and, since round 3, by reduced_top_mc right behind the last tree
//  level -- the workgroup of a problem's last three levels goes straight on to that problem's top-down sweep)
// (bp, zsep: multiple right-hand sides per problem, rb_forward_top<.., MULTI> -- the records are those of problem bp, the
//  z_sep of this right-hand side live in zsep[b][N][NX])
template <int NX>
__device__ __forceinline__ void backsub_top_body(const Dims& d, const int b, const int t,
                                                 const double* __restrict__ recs, double* __restrict__ ytop,
                                                 double* ytop_lds, const int bp, const double* zsep) {
  constexpr int NN = NX * NX, REC = 2 * NN + NX;
  const int N = d.N, K = d.K;
  const int br = bp >= 0 ? bp : b;
  for (int L = K - 1; L >= 3; --L) {
    const int T = 2 << L, nsep = N >> (L + 1);
    for (int it = t; it < nsep * NX; it += 256) {
      const int q = it / NX, r = it - q * NX;
      const int base = q * T, s = base + (T >> 1) - 1;
      const double* rc = recs + ((size_t)br * N + s) * REC;
      double acc = zsep ? zsep[((size_t)b * N + s) * NX + r] : rc[2 * NN + r];
      if (base > 0) {
        double f[NX];
        load_row<NX>(rc + r * NX, f);
        const double* y = ytop_lds + ((base - 1) >> 3) * NX;
#pragma unroll
        for (int c = 0; c < NX; ++c) acc = fma(-f[c], y[c], acc);
      }
      if (base + T < N) {
        double f[NX];
        load_row<NX>(rc + NN + r * NX, f);
        const double* y = ytop_lds + ((base + T - 1) >> 3) * NX;
#pragma unroll
        for (int c = 0; c < NX; ++c) acc = fma(-f[c], y[c], acc);
      }
      ytop_lds[(s >> 3) * NX + r] = acc;
    }
    __syncthreads();
  }
  double* out = ytop + (size_t)b * (N >> 3) * NX;
  for (int e = t; e < ((N >> 3) - 1) * NX; e += 256) out[e] = ytop_lds[e];
}

template <int NX>
__global__ __launch_bounds__(256) void rb_backsub_top(Dims d, const double* __restrict__ recs, double* __restrict__ ytop) {
  extern __shared__ double ytop_lds[];  // [N / 8][NX]: y of separator 8 j + 7
  backsub_top_body<NX>(d, blockIdx.x, threadIdx.x, recs, ytop, ytop_lds);
}

//   rb_backsub: grid (N / 8, batch), block 256; N >= 8. A workgroup needs nothing of its neighbours' (the two multipliers
//   next to its subtree come from `ytop`): a step that wants a knot range alone launches the workgroups of that range
//   (Dims::xoff = the first one, a shorter grid: NDLQR_SOLN_ONLY, launch_small.hpp).
// Everything the workgroup needs ([A | B] of its eight knots, its seven records, weights, right-hand
// sides: ~27 KB) arrives in ONE round of coalesced 16-byte loads and is staged in LDS; rows and columns
// are read from there. (Row- and column-wise global loads per thread made the address units the
// bottleneck of the first version: ~6 line requests per line actually fetched.)
template <int NX, int NU>
struct alignas(16) RbBacksubLds {
  static constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, REC = 2 * NN + NX, R0 = NX * (NX + 1) / 2;
  static constexpr int R0P = (R0 + 1) / 2 * 2;      // compact record, padded to whole 16-byte words
  static constexpr int WP = RbKnot<NX, NU>::WP;
  double ab[8][NX * WP];
  double rec0[4][R0P];    // level-0 separators first + 0, 2, 4, 6: Cholesky factor of S-bar, packed lower triangle
  double rec1[3][REC + (REC & 1)];  // separators first + 1, first + 3, first + 5: f_a | f_bb | z_sep
  double ys[9][NX];       // [0..6]: separators first .. first + 6; [7]: first - 1; [8]: first + 7
  double qs[8][W];        // 1 / [Q | R]
  double rs[8][ROWS];     // raw right-hand sides
  double dots[8][NX];     // A_i' y_i of the odd knots (state rows)
  double vs[4][NX];       // v of the four level-0 separators
};

// MULTI (several right-hand sides per problem, kernels below): grid.y counts right-hand sides, b % nprob is the problem whose
// inputs and records are read, z_sep of the level-1 / 2 separators come from zsep[b][N][NX] instead of the records.
template <int NX, int NU, bool MULTI = false>
__global__ __launch_bounds__(256) void rb_backsub(Dims d, const double* __restrict__ AB, const double* __restrict__ QR,
                                                  const double* __restrict__ rhs, const double* __restrict__ recs,
                                                  const double* __restrict__ ytop, double* __restrict__ z,
                                                  const int nprob = 0, const double* __restrict__ zsep = nullptr) {
  using Lds = RbBacksubLds<NX, NU>;
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, REC = 2 * NN + NX, KPB = 8, WP = Lds::WP;
  constexpr int R0 = Lds::R0;
  static_assert(9 * NX <= 256 && KPB * ROWS <= 256, "thread roles fit the workgroup");
  __shared__ Lds lds;
  const int N = d.N, b = blockIdx.y, first = (blockIdx.x + d.xoff) * KPB;
  const int bp = MULTI ? b % nprob : b;  // the problem whose inputs and records this right-hand side is solved against
  auto sep_slot = [&](int s) -> int { return s < first ? 7 : (s >= first + 7 ? 8 : s - first); };
  const int t = threadIdx.x;

  // ---- one round of loads
  {
    const double* abm = AB + ((size_t)bp * N + first) * NX * W;
    const double* rc0 = recs + ((size_t)bp * N + first) * REC;
    // 16-byte loads wherever the pieces are 16-byte granules: the eight knots' [A | B] always (8 NX W doubles from
    // a tile boundary), the records when NX is even (REC and R0 even)
    constexpr int NA = KPB * NX * W / 2, IA = (NA + 255) / 256;                // [A | B] in 16-byte words
    constexpr bool WIDE = NX % 2 == 0 && R0 % 2 == 0;
    constexpr int G = WIDE ? 2 : 1;                                             // doubles per record load
    constexpr int N0 = 4 * R0 / G, I0 = (N0 + 255) / 256, N1 = 3 * REC / G, I1 = (N1 + 255) / 256;
    constexpr int NQ = KPB * W, NR = KPB * ROWS;
    static_assert(NQ <= 256 && NR <= 256, "one load per thread for the vectors");
    static_assert((KPB * NX * W) % 2 == 0, "eight knots of [A | B] are whole 16-byte words");
    // [A | B]: where a knot is a whole number of 32-thread rounds of 16-byte words (96 at (12,4)), thread t takes words
    // (t & 31) + 32 j of knot t >> 5 -- the same coalescing, and no division by the knot size in the index arithmetic
    // (the constant divisions of the staging indices were a sixth of this kernel's vector instructions)
    constexpr int KW = NX * W / 2;                                             // 16-byte words per knot (W even)
    constexpr bool BYKNOT = (NX * W) % 2 == 0 && KW % 32 == 0;
    constexpr int JA = BYKNOT ? KW / 32 : IA;
    double2 ta[JA];
    double t0[I0][G], t1[I1][G];
#pragma unroll
    for (int it = 0; it < JA; ++it) {
      if constexpr (BYKNOT) {
        ta[it] = reinterpret_cast<const double2*>(abm)[(t >> 5) * KW + (t & 31) + 32 * it];
      } else {
        const int e = t + 256 * it;
        ta[it] = reinterpret_cast<const double2*>(abm)[e < NA ? e : NA - 1];
      }
    }
#pragma unroll
    for (int it = 0; it < I0; ++it) {
      const int e = t + 256 * it, ec = e < N0 ? e : N0 - 1, j = ec / (R0 / G), w_ = ec - j * (R0 / G);
      const double* src = rc0 + (size_t)(2 * j) * REC + G * w_;
      if constexpr (WIDE) { const double2 v = *reinterpret_cast<const double2*>(src); t0[it][0] = v.x; t0[it][1] = v.y; }
      else t0[it][0] = *src;
    }
#pragma unroll
    for (int it = 0; it < I1; ++it) {
      const int e = t + 256 * it, ec = e < N1 ? e : N1 - 1, j = ec / (REC / G), w_ = ec - j * (REC / G);
      const double* src = rc0 + (size_t)(2 * j + 1) * REC + G * w_;
      if constexpr (WIDE) { const double2 v = *reinterpret_cast<const double2*>(src); t1[it][0] = v.x; t1[it][1] = v.y; }
      else t1[it][0] = *src;
    }
    const double tq = QR[((size_t)bp * N + first) * W + (t < NQ ? t : NQ - 1)];
    const double tr = rhs[((size_t)b * N + first) * ROWS + (t < NR ? t : NR - 1)];
    double tz = 0.0;
    if constexpr (MULTI) {  // z_sep of separators first + 1, + 3, + 5 of THIS right-hand side
      const int e = t < 3 * NX ? t : 3 * NX - 1, j = e / NX;
      tz = zsep[((size_t)b * N + first + 2 * j + 1) * NX + (e - j * NX)];
    }
    double ty = 0.0;
    if (t < 2 * NX) {  // the two multipliers next to the workgroup's subtree
      const int sx = t < NX ? first - 1 : first + 7;
      if (sx >= 0 && sx < N - 1) ty = ytop[((size_t)b * (N >> 3) + (sx >> 3)) * NX + (t < NX ? t : t - NX)];
    }
#pragma unroll
    for (int it = 0; it < JA; ++it) {
      if constexpr (BYKNOT) {
        const int w_ = 2 * ((t & 31) + 32 * it), row = w_ / W, c = w_ - row * W;  // (W even: both doubles in one row)
        *reinterpret_cast<double2*>(&lds.ab[t >> 5][row * WP + c]) = ta[it];
      } else {
        const int e = t + 256 * it, ec = e < NA ? e : NA - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // (the two doubles of a word may sit in different rows when W is odd)
          const int ed = 2 * ec + h, kn = ed / (NX * W), w_ = ed - kn * NX * W;
          const int row = w_ / W, c = w_ - row * W;
          lds.ab[kn][row * WP + c] = h ? ta[it].y : ta[it].x;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < I0; ++it) {
      const int e = t + 256 * it, ec = e < N0 ? e : N0 - 1, j = ec / (R0 / G), w_ = ec - j * (R0 / G);
#pragma unroll
      for (int h = 0; h < G; ++h) lds.rec0[j][G * w_ + h] = t0[it][h];
    }
#pragma unroll
    for (int it = 0; it < I1; ++it) {
      const int e = t + 256 * it, ec = e < N1 ? e : N1 - 1, j = ec / (REC / G), w_ = ec - j * (REC / G);
#pragma unroll
      for (int h = 0; h < G; ++h) lds.rec1[j][G * w_ + h] = t1[it][h];
    }
    (&lds.qs[0][0])[t < NQ ? t : NQ - 1] = 1.0 / tq;
    (&lds.rs[0][0])[t < NR ? t : NR - 1] = tr;
    if (t < 2 * NX) lds.ys[t < NX ? 7 : 8][t < NX ? t : t - NX] = ty;
    if constexpr (MULTI) {
      __syncthreads();  // (the staged records carry the z_sep of whatever right-hand side was solved last)
      if (t < 3 * NX) lds.rec1[t / NX][2 * NN + t % NX] = tz;
    }
  }
  __syncthreads();

  const int q = t / NX, r = t - q * NX;
  const bool sep_thread = q < 7;
  const int s = first + (sep_thread ? q : 0), l = trailing_ones(s);  // l <= 2 for the local separators
  const int sbase = s - ((1 << l) - 1);
  const bool hasA = sbase > 0, hasB = sbase + (2 << l) < N;
  const int slotA = hasA ? sep_slot(sbase - 1) : 0, slotB = hasB ? sep_slot(sbase + (2 << l) - 1) : 0;
  const int kn = t / ROWS, rr = t - kn * ROWS;  // output role: knot kn of the workgroup, row rr
  const bool out_thread = kn < KPB;
  const int kc = out_thread ? kn : 0;
  const int i = first + kc;
  const bool lam = rr < NX;
  const int col = lam ? rr : rr - NX;  // column of [A_i | B_i] this row dots with y_i
  const bool needs_ab = out_thread && i < N - 1 && (lam ? i == 0 : !(i == 0 && rr < 2 * NX));
  auto ab_dot = [&](const double* y) {  // column `col` of [A_i | B_i] times y
    const double* am = lds.ab[kc] + col;
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < NX; ++c) a = fma(am[c * WP], y[c], a);
    return a;
  };

  // ---- multipliers of the local separators of level 2 and 1
  for (int Lv = 2; Lv >= 1; --Lv) {
    if (sep_thread && l == Lv) {
      const double* rc = lds.rec1[q >> 1];
      double a = rc[2 * NN + r];
      if (hasA) {
#pragma unroll
        for (int c = 0; c < NX; ++c) a = fma(-rc[r * NX + c], lds.ys[slotA][c], a);
      }
      if (hasB) {
#pragma unroll
        for (int c = 0; c < NX; ++c) a = fma(-rc[NN + r * NX + c], lds.ys[slotB][c], a);
      }
      lds.ys[q][r] = a;
    }
    __syncthreads();
  }
  // ---- A_i' y_i of the odd knots (their separators are of level >= 1): part of v and of x_i
  const bool odd_state = out_thread && (kn & 1) && rr >= NX && rr < 2 * NX;
  double dot = 0.0;
  if (odd_state) {
    if (needs_ab) dot = ab_dot(lds.ys[sep_slot(i)]);
    lds.dots[kn][rr - NX] = dot;
  }
  __syncthreads();
  // ---- level-0 separators: v, then y = L^-T L^-1 v
  if (sep_thread && l == 0) {
    const int k = s - first;  // even knot of the workgroup
    const bool fst = s == 0;
    const double* arow = lds.ab[k] + r * WP;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const double wq = lds.qs[k][c];
      double zh;  // z-hat(c) + what r_a y_{s-1} contributes through column c
      if (c < NX) zh = fst ? -lds.rs[k][c] : (lds.rs[k][NX + c] + (hasA ? lds.ys[slotA][c] : 0.0)) * wq;
      else zh = lds.rs[k][NX + c] * wq;
      v = fma(arow[c], zh, v);
    }
    const double q1 = lds.qs[k + 1][r];
    v -= lds.rs[k + 1][r] + (lds.rs[k + 1][NX + r] - lds.dots[k + 1][r]) * q1;
    lds.vs[k >> 1][r] = v;
  }
  __syncthreads();
  // the compact records hold the Cholesky factor L of S-bar (packed rows): y = L^-T (L^-1 v)
  if (t < 64) {
    // wavefront 0: the four level-0 separators of the tile in its four DPP rows, row r of L per lane; both
    // substitutions broadcast the freshly resolved entry inside the row (v_mov_b64_dpp row_newbcast): no LDS
    // round trip and no barrier per step. (All 64 lanes run it -- DPP needs them active --, lanes r >= NX idle.)
    const int j = t >> 4, r15 = t & 15, rc = r15 < NX ? r15 : NX - 1;
    const double* Lp = lds.rec0[j];
    const double dinv = 1.0 / Lp[rc * (rc + 1) / 2 + rc];
    double lrow[NX], lcol[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      const double lo = Lp[rc * (rc + 1) / 2 + (c < rc ? c : rc)];  // L(r, c), c < r
      const double up = Lp[(c > rc ? c : rc) * ((c > rc ? c : rc) + 1) / 2 + rc];  // L(c, r), c > r
      lrow[c] = (c < r15 && r15 < NX) ? lo : 0.0;
      lcol[c] = (c > r15 && r15 < NX) ? up : 0.0;
    }
    double x = lds.vs[j][rc];
    sfor<NX>([&](auto cc) {  // forward: t_c = x_c / L(c, c) is final when step c starts
      constexpr int c = decltype(cc)::value;
      double xs = x * dinv;
      dpp_fence(xs);
      x = fma(-lrow[c], row_bc<c>(xs), x);
    });
    x = x * dinv;  // t_r
    sfor<NX>([&](auto cc) {  // backward, c descending: y_c is final when its step starts
      constexpr int c = NX - 1 - decltype(cc)::value;
      double xs = x * dinv;
      dpp_fence(xs);
      x = fma(-lcol[c], row_bc<c>(xs), x);
    });
    if (r15 < NX) lds.ys[2 * j][r15] = x * dinv;
  }
  __syncthreads();

  // ---- solution rows
  if (!out_thread) return;
  const double rv = lds.rs[kn][rr];
  double out;
  const double* yi = lds.ys[sep_slot(i < N - 1 ? i : i - 1)];        // y_i   (unused for the last knot)
  const double* yp = lds.ys[sep_slot(i > 0 ? i - 1 : 0)];            // y_{i-1} (unused for knot 0)
  if (needs_ab && !odd_state) dot = ab_dot(yi);
  if (lam) {
    out = (i == 0) ? fma(-QR[(size_t)bp * N * W + rr], rv, -lds.rs[0][NX + rr]) + dot : yp[rr];
  } else if (rr < 2 * NX) {
    out = (i == 0) ? -lds.rs[0][rr - NX] : (rv - dot + yp[rr - NX]) * lds.qs[kn][rr - NX];
  } else {
    out = (i == N - 1) ? rv : (rv - dot) * lds.qs[kn][rr - NX];
  }
  z[((size_t)b * N + i) * ROWS + rr] = out;
}

// ------------------------------------------------------------------------------------- right-hand-side re-solve (round 4)
// New q, r, d, x0 against what a solve of the default schedule with NDLQR_FLAG_KEEP_RECORDS leaves behind:
// the compact level-0 records (L of S-bar, packed), the records f_a | f_bb of every separator of level >= 1 and --
// kept for this -- its Cholesky factor L, n x n row-major, in the slack of the record slot in front of it (slot s - 1
// belongs to a level-0 separator, which uses n (n + 1) / 2 of its 2 n^2 + n doubles: rb_lrec). Forward pass over the
// separators, the right-hand-side column of the factorisation alone (b~ = leaf - gL - gR):
//     level 0:    z_s = (L L')^-1 leafb_s;   gR[s - 1] += r_a(s)' z_s,  gL[s + 1] += r_bb(s)' z_s   (couplings from the data)
//     level >= 1: b~_s = leafb_s - gL[s] - gR[s];  z_sep(s) = (L L')^-1 b~_s -> record;
//                 gR[A] += f_a(s)' b~_s,  gL[B] += f_bb(s)' b~_s        (r_a' S-bar^-1 b~ = (S-bar^-1 r_a)' b~)
// then the back-substitution of a full solve (backsub_top_body, rb_backsub) as it stands. Two launches:
//   rb_forward      grid (N / 8, batch), block 256: levels 0-2 of eight knots (the staging of rb_backsub plus the
//                   three factors of level 1 / 2); what the block adds to its two outer separators -> fsum
//   rb_forward_top  grid (batch), block 256, dynamic LDS 4 (N / 8) NX doubles: levels >= 3 of a problem, level by level,
//                   and straight on to the top-down sweep -> ytop
// against 0.73 ms of the full-record re-solve (rhs_forward_small / _upper, backsub_small) and 0.59 of a full solve.
// x <- (L L')^-1 x for the separator of this lane's 16-lane DPP row (lane r15 of the row: entry r15; lrow[c] = L(r15, c)
// for c < r15, lcol[c] = L(c, r15) for c > r15, zero elsewhere; dinv = 1 / L(r15, r15)). Every lane of the wavefront
// executes it (DPP); the freshly resolved entry is broadcast inside the row, no LDS round trip and no barrier per step.
template <int NX>
__device__ __forceinline__ double rb_llt_solve(double x, const double (&lrow)[NX], const double (&lcol)[NX], const double dinv) {
  sfor<NX>([&](auto cc) {  // forward: t_c = x_c / L(c, c) is final when step c starts
    constexpr int c = decltype(cc)::value;
    double xs = x * dinv;
    dpp_fence(xs);
    x = fma(-lrow[c], row_bc<c>(xs), x);
  });
  x = x * dinv;  // t_r
  sfor<NX>([&](auto cc) {  // backward, c descending: y_c is final when its step starts
    constexpr int c = NX - 1 - decltype(cc)::value;
    double xs = x * dinv;
    dpp_fence(xs);
    x = fma(-lcol[c], row_bc<c>(xs), x);
  });
  return x * dinv;
}

template <int NX, int NU>
struct alignas(16) RbForwardLds {
  static constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, REC = 2 * NN + NX, R0 = NX * (NX + 1) / 2;
  static constexpr int R0P = (R0 + 1) / 2 * 2, WP = RbKnot<NX, NU>::WP;
  double ab[8][NX * WP];
  double rec0[4][R0P];    // level-0 separators first + 0, 2, 4, 6: L, packed lower triangle
  double l1[3][NN];       // separators first + 1, first + 3, first + 5: L, n x n row-major
  double f1[3][2 * NN];   // ... and their records f_a | f_bb
  double qs[8][W];        // 1 / [Q | R]
  double rs[8][ROWS];     // raw right-hand sides
  double bt[7][NX];       // b~ of the local separators (level 0: leafb)
  double zs[4][NX];       // z of the level-0 separators
  double gl[7][NX], gr[7][NX];  // what has been pushed to the local separators of level 1, 2 (slots 1, 3, 5)
  double outl[3][NX], outr[3][NX];  // contributions to first - 1 (from first, first + 1, first + 3) / first + 7 (first + 6, + 5, + 3)
};

template <int NX, int NU, bool MULTI = false>
__global__ __launch_bounds__(256) void rb_forward(Dims d, const double* __restrict__ AB, const double* __restrict__ QR,
                                                  const double* __restrict__ rhs, double* __restrict__ recs,
                                                  double* __restrict__ fsum, const int nprob = 0,
                                                  double* __restrict__ zsep = nullptr) {
  using Lds = RbForwardLds<NX, NU>;
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, REC = 2 * NN + NX, KPB = 8, WP = Lds::WP, R0 = Lds::R0;
  static_assert(7 * NX <= 256 && KPB * ROWS <= 256 && KPB * W <= 256, "thread roles fit the workgroup");
  __shared__ Lds lds;
  const int N = d.N, b = blockIdx.y, first = blockIdx.x * KPB, t = threadIdx.x;
  const int bp = MULTI ? b % nprob : b;  // (MULTI: see rb_backsub)
  // ---- staging: [A | B] of the eight knots, weights, right-hand sides, the seven separators' factors and records --
  //      every load is requested before the first store (16-byte words wherever the pieces are 16-byte granules, like
  //      rb_backsub's)
  {
    const double* abm = AB + ((size_t)bp * N + first) * NX * W;
    const double* rc0 = recs + ((size_t)bp * N + first) * REC;
    constexpr bool WIDE = NX % 2 == 0 && R0 % 2 == 0 && (NX * W) % 2 == 0;
    constexpr int G = WIDE ? 2 : 1;
    constexpr int NA = KPB * NX * W / G, IA = (NA + 255) / 256;
    constexpr int N0 = 4 * R0 / G, I0 = (N0 + 255) / 256, NL = 3 * NN / G, IL = (NL + 255) / 256;
    constexpr int NF = 3 * 2 * NN / G, IF = (NF + 255) / 256;
    double ta[IA][G], t0[I0][G], tl[IL][G], tf[IF][G];
    auto ld = [](const double* src, double (&dst)[G]) {
      if constexpr (WIDE) { const double2 v = *reinterpret_cast<const double2*>(src); dst[0] = v.x; dst[1] = v.y; }
      else dst[0] = *src;
    };
#pragma unroll
    for (int it = 0; it < IA; ++it) { const int e = t + 256 * it, ec = e < NA ? e : NA - 1; ld(abm + G * ec, ta[it]); }
#pragma unroll
    for (int it = 0; it < I0; ++it) {
      const int e = t + 256 * it, ec = e < N0 ? e : N0 - 1, j = ec / (R0 / G), w_ = ec - j * (R0 / G);
      ld(rc0 + (size_t)(2 * j) * REC + G * w_, t0[it]);
    }
#pragma unroll
    for (int it = 0; it < IL; ++it) {  // (factor of first + 2 j + 1: slack of slot first + 2 j)
      const int e = t + 256 * it, ec = e < NL ? e : NL - 1, j = ec / (NN / G), w_ = ec - j * (NN / G);
      ld(rc0 + (size_t)(2 * j) * REC + rb_lrec_offset<NX>() + G * w_, tl[it]);
    }
#pragma unroll
    for (int it = 0; it < IF; ++it) {
      const int e = t + 256 * it, ec = e < NF ? e : NF - 1, j = ec / (2 * NN / G), w_ = ec - j * (2 * NN / G);
      ld(rc0 + (size_t)(2 * j + 1) * REC + G * w_, tf[it]);
    }
    const double tq = QR[((size_t)bp * N + first) * W + (t < KPB * W ? t : KPB * W - 1)];
    const double tr = rhs[((size_t)b * N + first) * ROWS + (t < KPB * ROWS ? t : KPB * ROWS - 1)];
#pragma unroll
    for (int it = 0; it < IA; ++it) {
      const int e = t + 256 * it, ec = e < NA ? e : NA - 1;
#pragma unroll
      for (int h = 0; h < G; ++h) {
        const int ed = G * ec + h, kn = ed / (NX * W), w_ = ed - kn * NX * W, row = w_ / W, c = w_ - row * W;
        lds.ab[kn][row * WP + c] = ta[it][h];
      }
    }
#pragma unroll
    for (int it = 0; it < I0; ++it) {
      const int e = t + 256 * it, ec = e < N0 ? e : N0 - 1, j = ec / (R0 / G), w_ = ec - j * (R0 / G);
#pragma unroll
      for (int h = 0; h < G; ++h) lds.rec0[j][G * w_ + h] = t0[it][h];
    }
#pragma unroll
    for (int it = 0; it < IL; ++it) {
      const int e = t + 256 * it, ec = e < NL ? e : NL - 1;
#pragma unroll
      for (int h = 0; h < G; ++h) (&lds.l1[0][0])[G * ec + h] = tl[it][h];
    }
#pragma unroll
    for (int it = 0; it < IF; ++it) {
      const int e = t + 256 * it, ec = e < NF ? e : NF - 1;
#pragma unroll
      for (int h = 0; h < G; ++h) (&lds.f1[0][0])[G * ec + h] = tf[it][h];
    }
    (&lds.qs[0][0])[t < KPB * W ? t : KPB * W - 1] = 1.0 / tq;
    (&lds.rs[0][0])[t < KPB * ROWS ? t : KPB * ROWS - 1] = tr;
  }
  __syncthreads();
  const int q = t / NX, r = t - q * NX;
  const bool sep_thread = q < 7;
  // ---- leafb of the seven local separators: [A_s | B_s] z-hat(s) - z(s+1).lambda - z(s+1).x / Q_{s+1}
  if (sep_thread) {
    const int s = first + q;
    const bool fst = s == 0;
    const double* arow = lds.ab[q] + r * WP;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const double wq = lds.qs[q][c];
      const double zh = (c < NX && fst) ? -lds.rs[q][c] : lds.rs[q][NX + c] * wq;
      v = fma(arow[c], zh, v);
    }
    v -= lds.rs[q + 1][r] + lds.rs[q + 1][NX + r] * lds.qs[q + 1][r];
    lds.bt[q][r] = v;
  }
  __syncthreads();
  // one wavefront solves with up to four factors at once, one per 16-lane DPP row: sel(j) = local separator of row j
  auto solve_rows = [&](const int nrows, auto sel, const bool packed) {
    if (t < 64) {
      const int j = t >> 4, r15 = t & 15, rc = r15 < NX ? r15 : NX - 1;
      const int jj = j < nrows ? j : nrows - 1;  // (idle rows shadow a live one)
      const int sl = sel(jj);
      double lrow[NX], lcol[NX], dinv;
      if (packed) {
        const double* Lp = lds.rec0[sl >> 1];
        dinv = 1.0 / Lp[rc * (rc + 1) / 2 + rc];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          const double lo = Lp[rc * (rc + 1) / 2 + (c < rc ? c : rc)];
          const double up = Lp[(c > rc ? c : rc) * ((c > rc ? c : rc) + 1) / 2 + rc];
          lrow[c] = (c < r15 && r15 < NX) ? lo : 0.0;
          lcol[c] = (c > r15 && r15 < NX) ? up : 0.0;
        }
      } else {
        const double* Lf = lds.l1[sl >> 1];
        dinv = 1.0 / Lf[rc * NX + rc];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
          lrow[c] = (c < r15 && r15 < NX) ? Lf[rc * NX + c] : 0.0;
          lcol[c] = (c > r15 && r15 < NX) ? Lf[c * NX + rc] : 0.0;
        }
      }
      const double x = rb_llt_solve<NX>(lds.bt[sl][rc], lrow, lcol, dinv);
      if (r15 < NX && j < nrows) {
        if (packed) lds.zs[sl >> 1][r15] = x;
        else if constexpr (MULTI) zsep[((size_t)b * N + first + sl) * NX + r15] = x;
        else recs[((size_t)b * N + first + sl) * REC + 2 * NN + r15] = x;  // z_sep of a separator of level 1 / 2
      }
    }
  };
  // ---- level 0
  solve_rows(4, [](int j) { return 2 * j; }, true);
  __syncthreads();
  // what the level-0 separators push: s = first + 2 j to s - 1 (gR) and s + 1 (gL); threads (j, side, r)
  if (t < 8 * NX) {
    const int j = t / (2 * NX), side = (t / NX) & 1, rr = t % NX, k = 2 * j;
    double a = 0.0;
    if (side == 0) {  // gR[s - 1](rr) = -(1 / Q_s(rr)) (A_s' z)(rr)
      if (first + k > 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) a = fma(lds.ab[k][i * WP + rr], lds.zs[j][i], a);
        a = -a * lds.qs[k][rr];
      }
      if (j == 0) lds.outl[0][rr] = a; else lds.gr[k - 1][rr] = a;
    } else {          // gL[s + 1](rr) = -(A_{s+1} (z / Q_{s+1}))(rr)
#pragma unroll
      for (int i = 0; i < NX; ++i) a = fma(lds.ab[k + 1][rr * WP + i], lds.zs[j][i] * lds.qs[k + 1][i], a);
      a = -a;
      if (j == 3) lds.outr[0][rr] = a; else lds.gl[k + 1][rr] = a;
    }
  }
  __syncthreads();
  // ---- level 1: t = first + 1, first + 5 (local 1, 5)
  if (t < 2 * NX) { const int sl = t < NX ? 1 : 5, rr = t % NX; lds.bt[sl][rr] -= lds.gl[sl][rr] + lds.gr[sl][rr]; }
  __syncthreads();
  solve_rows(2, [](int j) { return 4 * j + 1; }, false);
  // their pushes: f_a' b~ to the separator left of the subtree, f_bb' b~ to the one right of it; threads (which, side, c)
  if (t >= 64 && t < 64 + 4 * NX) {
    const int u = t - 64, which = u / (2 * NX), side = (u / NX) & 1, c = u % NX, sl = 4 * which + 1;
    const double* f = lds.f1[sl >> 1] + side * NN;
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) a = fma(f[i * NX + c], lds.bt[sl][i], a);
    // local 1: A = first - 1, B = local 3;   local 5: A = local 3, B = first + 7
    if (which == 0) { if (side == 0) lds.outl[1][c] = (first > 0) ? a : 0.0; else lds.gl[3][c] += a; }
    else { if (side == 0) lds.gr[3][c] += a; else lds.outr[1][c] = (first + 8 < N) ? a : 0.0; }
  }
  __syncthreads();
  // ---- level 2: local 3
  if (t < NX) lds.bt[3][t] -= lds.gl[3][t] + lds.gr[3][t];
  __syncthreads();
  solve_rows(1, [](int) { return 3; }, false);
  if (t >= 64 && t < 64 + 2 * NX) {
    const int u = t - 64, side = u / NX, c = u % NX;
    const double* f = lds.f1[1] + side * NN;
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) a = fma(f[i * NX + c], lds.bt[3][i], a);
    if (side == 0) lds.outl[2][c] = (first > 0) ? a : 0.0; else lds.outr[2][c] = (first + 8 < N) ? a : 0.0;
  }
  __syncthreads();
  // ---- what this block adds to gR of separator first - 1 and to gL of separator first + 7
  if (t < 2 * NX) {
    const int side = t / NX, c = t % NX;
    const double v = side == 0 ? lds.outl[0][c] + lds.outl[1][c] + lds.outl[2][c] : lds.outr[0][c] + lds.outr[1][c] + lds.outr[2][c];
    fsum[(((size_t)b * (N >> 3) + blockIdx.x) * 2 + side) * NX + c] = v;
  }
}

// levels >= 3 of one problem + the top-down sweep. LDS: bt | gl | gr | ytop, each [N / 8][NX] (entry m: separator 8 m + 7).
template <int NX, int NU, bool MULTI = false>
__global__ __launch_bounds__(256) void rb_forward_top(Dims d, const double* __restrict__ AB, const double* __restrict__ QR,
                                                      const double* __restrict__ rhs, double* __restrict__ recs,
                                                      const double* __restrict__ fsum, double* __restrict__ ytop,
                                                      const int nprob = 0, double* __restrict__ zsep = nullptr) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, REC = 2 * NN + NX;
  extern __shared__ double sm[];
  const int N = d.N, K = d.K, b = blockIdx.x, t = threadIdx.x, M = N >> 3;
  const int bp = MULTI ? b % nprob : b;  // (MULTI: see rb_backsub)
  double* bt = sm;
  double* gl = bt + M * NX;
  double* gr = gl + M * NX;
  double* yt = gr + M * NX;
  // ---- leafb of every separator 8 m + 7 minus what the blocks of rb_forward pushed to it (block m: gL, block m + 1: gR)
  for (int e = t; e < (M - 1) * NX; e += 256) {
    const int mq = e / NX, r = e - mq * NX, s = 8 * mq + 7;
    const double* arow = AB + (((size_t)bp * N + s) * NX + r) * W;
    const double* qr = QR + ((size_t)bp * N + s) * W;
    const double* r0 = rhs + ((size_t)b * N + s) * ROWS;
    double v = 0.0;
    for (int c = 0; c < W; ++c) v = fma(arow[c], r0[NX + c] / qr[c], v);  // (s >= 7: never the first knot)
    v -= r0[ROWS + r] + r0[ROWS + NX + r] / qr[W + r];
    v -= fsum[(((size_t)b * M + mq) * 2 + 1) * NX + r] + fsum[(((size_t)b * M + mq + 1) * 2 + 0) * NX + r];
    bt[e] = v;
    gl[e] = 0.0;
    gr[e] = 0.0;
  }
  __syncthreads();
  for (int L = 3; L < K; ++L) {
    const int T = 2 << L, nsep = N >> (L + 1);
    // b~ and z_sep = (L L')^-1 b~ of the level's separators, one per 16-lane DPP row, sixteen per round
    for (int q0 = 0; q0 < nsep; q0 += 16) {
      const int j = t >> 4, r15 = t & 15, rc = r15 < NX ? r15 : NX - 1;
      const int qq = q0 + j < nsep ? q0 + j : nsep - 1;
      const int s = qq * T + (T >> 1) - 1, mq = s >> 3;
      const double* Lf = rb_lrec<NX>((const double*)recs, d, bp, s);
      double lrow[NX], lcol[NX];
      const double dinv = 1.0 / Lf[rc * NX + rc];
#pragma unroll
      for (int c = 0; c < NX; ++c) {
        lrow[c] = (c < r15 && r15 < NX) ? Lf[rc * NX + c] : 0.0;
        lcol[c] = (c > r15 && r15 < NX) ? Lf[c * NX + rc] : 0.0;
      }
      const double bv = bt[mq * NX + rc] - gl[mq * NX + rc] - gr[mq * NX + rc];
      const double x = rb_llt_solve<NX>(bv, lrow, lcol, dinv);
      if (r15 < NX && q0 + j < nsep) {
        bt[mq * NX + r15] = bv;
        if constexpr (MULTI) zsep[((size_t)b * N + s) * NX + r15] = x;
        else recs[((size_t)b * N + s) * REC + 2 * NN + r15] = x;
      }
    }
    __syncthreads();
    // pushes of the level: f_a' b~ -> gR of the separator left of the subtree, f_bb' b~ -> gL of the one right of it
    for (int e = t; e < nsep * 2 * NX; e += 256) {
      const int qq = e / (2 * NX), side = (e / NX) & 1, c = e % NX;
      const int base = qq * T, s = base + (T >> 1) - 1;
      const int nb = side == 0 ? base - 1 : base + T - 1;
      if (nb < 0 || nb >= N - 1) continue;
      const double* f = recs + ((size_t)bp * N + s) * REC + side * NN;
      const double* bs = bt + (s >> 3) * NX;
      double a = 0.0;
#pragma unroll
      for (int i = 0; i < NX; ++i) a = fma(f[i * NX + c], bs[i], a);
      (side == 0 ? gr : gl)[(nb >> 3) * NX + c] += a;  // (one contributor per level and side: no race)
    }
    __syncthreads();
  }
  if constexpr (MULTI) backsub_top_body<NX>(d, b, t, recs, ytop, yt, bp, zsep);
  else backsub_top_body<NX>(d, b, t, recs, ytop, yt);
}


}  // namespace ndlqr
