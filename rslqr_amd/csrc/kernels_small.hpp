// kernels_small.hpp -- size-specialised kernels for small (nstates, ninputs): one factor-block
// ROW per lane, two knots per 64-wide wavefront, separator math by one wavefront per separator.
//
// Same arithmetic (and, with STRICT, the same operation order) as kernels_generic.hpp; see that
// file for the mapping to the reference functions and DESIGN.md section 2 for the schedule:
//
//   bottom_small       leaf phase + tree levels 0..JB-1 fused on chip (registers + LDS exchange)
//   separator_core     S-bar, f_a, f_bb, z_sep of one separator by one wavefront (device function;
//                      products on the matrix cores in fast mode)
//   level_small        one upper level: separator (separator_wave) + Schur update of the first and
//                      last knot of every subtree (schur_rows), one wavefront per subtree
//   backsub_small      fast mode without KEEP: solution by back-substitution over the separator
//                      records and the problem data
//   apply_small        strict / KEEP: every knot through all upper levels in registers, one pass
//   rhs_forward_small, rhs_forward_upper
//                      new right-hand side against cached records + factors (then backsub_small)
//
// Variants that were measured and dropped (numbers in DESIGN.md, "Tried and dropped"): two separators per
// wavefront with L broadcast from LDS, an LDS-resident Cholesky/substitution with rolled pivot
// loops, and an apply kernel with two rows per lane -- each lost to latency at low occupancy.
#pragma once
#include "kernels_common.hpp"

namespace ndlqr {



// One factor-block row (NX doubles, 16-byte aligned when NX is even) <-> registers, as
// 16-byte vector accesses (global_load/store_dwordx4).
template <int NX>
__device__ __forceinline__ void load_row(const double* __restrict__ p, double (&v)[NX]) {
  if constexpr (NX % 2 == 0) {
    const double2* p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int k = 0; k < NX / 2; ++k) { const double2 t = p2[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
  } else {
#pragma unroll
    for (int k = 0; k < NX; ++k) v[k] = p[k];
  }
}
template <int NX>
__device__ __forceinline__ void store_row(double* __restrict__ p, const double (&v)[NX]) {
  if constexpr (NX % 2 == 0) {
    double2* p2 = reinterpret_cast<double2*>(p);
#pragma unroll
    for (int k = 0; k < NX / 2; ++k) p2[k] = make_double2(v[2 * k], v[2 * k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < NX; ++k) p[k] = v[k];
  }
}

// ------------------------------------------------------------------------------------- separator core
// Shared by separator_one and bottom_small: the level-l separator of one subtree computed by ONE
// wavefront from operands already in LDS. Written for instruction count: no data-dependent
// branches (a non-positive pivot only raises a flag; NaNs then propagate like in the reference),
// the 2 NX + 1 right-hand-side columns live in a 32-wide LDS panel so that no lane needs a
// select; L stays in the registers of lanes 0..NX-1 and is broadcast with v_readlane.
template <int NX, int NU>
struct alignas(16) SepIn {      // what a separator reads from its two neighbours
  double Exu[(NX + NU) * NX];   // state+input rows of E(s)
  double Axu[(NX + NU) * NX];   // state+input rows of the left outer column of knot s
  double E1x[NX * NX];          // state rows of E(s+1)
  double B1x[NX * NX];          // state rows of the right outer column of knot s+1
  double zxu[NX + NU];          // z(s) state+input
  double z1[2 * NX];            // z(s+1) lambda | state
};
template <int NX>
struct alignas(16) SepOut {
  static constexpr int NC = 32;  // panel columns: [f_a (NX) | f_bb (NX) | z_sep | unused]
  // row pitch: two doubles of padding put the rows that the NX lanes of a group write (P1) and
  // read (record stores) side by side into distinct LDS banks (32 doubles = all in one bank)
  static constexpr int LD = NC + 2;
  double X[NX * LD];             // right-hand sides in, solutions out; row k, column c
  double rdiag[NX];              // fast mode: 1 / L(j,j)
};

// The core is run by ONE whole wavefront; its LDS traffic is only ordered within that wavefront
// (wave_lds_sync), the caller provides the workgroup barriers around it.
// ab: row gi = lane % NX of [A_s | B_s], loaded by the caller (early, so that the latency of that
// load is not on the separator's critical path). Lrow: on return, row gi of the factor for
// lanes < NX (KEEPL: entries above the diagonal keep their S-bar values, like the reference's
// in-place factorisation). Returns true when a pivot was not positive. The caller issues the
// workgroup barrier that makes the solved panel visible to other wavefronts.
// Cholesky of S-bar (rows in acc, lanes 0..NX-1) and both substitutions of the 2 NX + 1 panel
// columns: the second half of a separator, shared by separator_core (products formed from the two
// neighbouring knots) and reduced_level (products assembled from the separator-only accumulators).
// after_forward(x): called with the lane's forward-substituted column between the two sweeps.
struct NoHook { template <class T> __device__ __forceinline__ void operator()(T&) const {} };

template <int NX, bool STRICT, bool KEEPL, int SEGB, class Hook>
__device__ __forceinline__ bool factor_solve(const int lane, double (&acc)[NX], SepOut<NX>& out,
                                             double (&Lrow)[NX], double* lstore, Hook after_forward) {
  constexpr int LD = SepOut<NX>::LD;
  const int gi = lane % NX;
  SEG_INIT();
  // P2: left-looking Cholesky on the registers of group 0 (every lane runs it; rows of other
  // groups are don't-cares), row j broadcast with v_readlane; finished columns go to LDS.
  bool bad = false;
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    double v = acc[j];
#pragma unroll
    for (int k = 0; k < j; ++k) v = mad<STRICT>(-acc[k], readlane_f64(acc[k], j), v);
    if constexpr (KEEPL) { if (gi >= j) acc[j] = v; } else { acc[j] = v; }
    const double pivot = readlane_f64(acc[j], j);
    bad |= !(pivot > 0.0);  // no short-circuit: keeps the pivot loop one basic block
    if constexpr (STRICT) {
      const double root = sqrt(pivot);
      if constexpr (KEEPL) { if (gi >= j) acc[j] = acc[j] / root; } else { acc[j] = acc[j] / root; }
    } else {
      const double rinv = rsqrt(pivot);
      if constexpr (KEEPL) { if (gi >= j) acc[j] = acc[j] * rinv; } else { acc[j] = acc[j] * rinv; }
      if (lane == 0) out.rdiag[j] = rinv;
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the row-j broadcasts (SGPRs) local to their column
  }
#pragma unroll
  for (int j = 0; j < NX; ++j) Lrow[j] = acc[j];
  if (lstore && lane < NX) store_row<NX>(lstore + gi * NX, acc);
  wave_lds_sync();
  SEG(SEGB + 2);

  // P3: one right-hand-side column per lane (lanes >= LD repeat a column: same values)
  const int col = lane & (SepOut<NX>::NC - 1);
  double x[NX];
#pragma unroll
  for (int k = 0; k < NX; ++k) x[k] = out.X[k * LD + col];
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    if constexpr (STRICT) x[j] = x[j] / readlane_f64(Lrow[j], j); else x[j] = x[j] * out.rdiag[j];
#pragma unroll
    for (int r = j + 1; r < NX; ++r) x[r] = mad<STRICT>(-readlane_f64(Lrow[j], r), x[j], x[r]);
    __builtin_amdgcn_sched_barrier(0);
  }
  after_forward(x);  // x = L^-1 (right-hand side): what the separator-only schedule pushes upwards
#pragma unroll
  for (int j = NX - 1; j >= 0; --j) {
    if constexpr (STRICT) x[j] = x[j] / readlane_f64(Lrow[j], j); else x[j] = x[j] * out.rdiag[j];
#pragma unroll
    for (int r = 0; r < j; ++r) x[r] = mad<STRICT>(-readlane_f64(Lrow[r], j), x[j], x[r]);
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_lds_sync();
  if (lane < SepOut<NX>::NC) {
#pragma unroll
    for (int k = 0; k < NX; ++k) out.X[k * LD + col] = x[k];
  }
  wave_lds_sync();
  SEG(SEGB + 3);
  return bad;
}

// Products of the separator on the matrix cores (fast mode, 6 <= NX <= 15; NX + NU is padded
// with zeros to a multiple of 4):
// [S-bar | rhs_z] = [A_s | B_s] [Exu | zxu] - [E1x | z1], rhs_a = [A_s | B_s] Axu as two 16x16 tiles
// of v_mfma_f64_16x16x4_f64. fp64 MFMA has no rate advantage on MI355X; what this buys is LDS
// traffic: every operand element is read once per tile (~45 LDS instructions) instead of once
// per lane row (~130), and LDS is the busiest pipe of the fused kernel.
template <int NX, int NU>
struct P1OnMatrixCores {
  static constexpr bool value = NX >= 6 && NX + 1 <= 16;  // smaller blocks are mostly tile padding
};

template <int NX, int NU, bool STRICT, bool KEEPL, int SEGB = 0, class Hook = NoHook>
__device__ __forceinline__ bool separator_core(const int lane, const double (&ab)[NX + NU],
                                               const SepIn<NX, NU>& in, SepOut<NX>& out,
                                               double (&Lrow)[NX], const double* abmat = nullptr,
                                               const int abpitch = 0, double* lstore = nullptr,
                                               Hook after_forward = Hook()) {
  // lstore: where to put the Cholesky factor (row gi at lstore + gi * NX) when only a run-time
  // flag asks for it (KEEP_RECORDS) -- stored right after the factorisation, so that no
  // register copy of it has to live until the caller gets round to storing it
  constexpr int W = NX + NU, LD = SepOut<NX>::LD;
  static_assert(2 * NX + 1 <= SepOut<NX>::NC && 2 * NX <= 64, "panel too narrow");
  const int grp = lane / NX, gi = lane - grp * NX;
  SEG_INIT();

  double acc[NX];
  if constexpr (!STRICT && P1OnMatrixCores<NX, NU>::value) {
    typedef double acc4 __attribute__((ext_vector_type(4)));
    constexpr int KS = (W + 3) / 4, SP = NX + 2;  // k-steps; pitch of the transposition scratch
    const int li = lane & 15, lk = lane >> 4;
    const int ri = li < NX ? li : NX - 1;  // rows / columns >= NX are padding: any finite data
    double af[KS], b0[KS], b1[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int kk = 4 * q + lk, k = kk < W ? kk : W - 1;  // k >= W: zero padding of the last step
      const bool kin = kk < W;
      af[q] = kin ? abmat[ri * abpitch + k] : 0.0;
      b0[q] = !kin ? 0.0 : (li < NX ? in.Exu[k * NX + li] : (li == NX ? in.zxu[k] : 0.0));
      b1[q] = (kin && li < NX) ? in.Axu[k * NX + li] : 0.0;
    }
    acc4 c0, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = lk + 4 * g, ic = i < NX ? i : NX - 1;
      c0[g] = li < NX ? -in.E1x[ic * NX + ri] : (li == NX ? -in.z1[ic] - in.z1[NX + ic] : 0.0);
    }
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], b0[q], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], b1[q], c1, 0, 0, 0);
    }
    // S-bar goes through a scratch over the consumed operands to get one row per lane; the
    // right-hand sides go straight to the panel
    double* scr = const_cast<double*>(in.Exu);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int i = lk + 4 * g;
      if (i < NX) {
        if (li < NX) { scr[i * SP + li] = c0[g]; out.X[i * LD + li] = c1[g]; }
        else if (li == NX) out.X[i * LD + 2 * NX] = c0[g];
      }
    }
    for (int e = lane; e < NX * NX; e += 64) {
      const int i = e / NX, j = e - i * NX;
      out.X[i * LD + NX + j] = -in.B1x[e];  // f_bb = -(state rows of F(s+1, bb))
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = scr[gi * SP + j];
  } else {
  // P1: row gi of S-bar (group 0) / of f_a (group 1)
#pragma unroll
  for (int j = 0; j < NX; ++j) acc[j] = 0.0;
  const double* M = (grp == 1) ? in.Axu : in.Exu;
#pragma unroll
  for (int k = 0; k < W; ++k) {
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = mad<STRICT>(ab[k], M[k * NX + j], acc[j]);
  }
  if (grp == 0) {
    double accz = -in.z1[gi];
#pragma unroll
    for (int k = 0; k < W; ++k) accz = mad<STRICT>(ab[k], in.zxu[k], accz);
    out.X[gi * LD + 2 * NX] = accz - in.z1[NX + gi];
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = acc[j] - in.E1x[gi * NX + j];
  } else if (grp == 1) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      out.X[gi * LD + j] = acc[j];
      out.X[gi * LD + NX + j] = -in.B1x[gi * NX + j];  // f_bb = -(state rows of F(s+1, bb))
    }
  }
  }  // vector-ALU products

  SEG(SEGB + 1);
  return factor_solve<NX, STRICT, KEEPL, SEGB>(lane, acc, out, Lrow, lstore, after_forward);
}

// One wavefront per separator: stage the operands (whole rows, 16-byte loads), run the core,
// store the record f_a | f_bb | z_sep and the lambda rows of knot s+1. `in` / `out` belong to the
// calling wavefront alone (all ordering is wave-local); on return the solved panel is in out.X.
// LAMBDA_OUT: also write the lambda rows of knot s+1 (its f_a / f_bb rows and the Cholesky
// factor) into F. A later full-level Schur pass or a factor download reads them; the
// boundary-first schedule takes them from the record instead.
template <int NX, int NU, bool STRICT, bool KEEP, bool LAMBDA_OUT>
__device__ __forceinline__ void separator_wave(const Dims& d, const int l, const int sub, const int b,
                                               const int lane, const double* __restrict__ AB, double* F,
                                               double* z, double* __restrict__ rec, int* __restrict__ info,
                                               SepIn<NX, NU>& in, SepOut<NX>& out,
                                               const bool store_l = false) {
  // store_l: write the Cholesky factor of this separator (lambda rows of knot s+1, column l) even
  // without LAMBDA_OUT -- what a record-based right-hand-side re-solve needs (KEEP_RECORDS)
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, LD = SepOut<NX>::LD;
  const int N = d.N;
  const int half = 1 << l, base = sub * (2 << l), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  SEG_INIT();
  const int gi = lane % NX;
  double ab[W];  // every lane loads a (valid) row: no exec-masked branches around the loads
  load_row<W>(AB + (((size_t)b * N + s) * NX + gi) * W, ab);
  {
    const double* Es = Fblk(F, d, b, l, s) + NN;
    const double* Fas = Fblk(F, d, b, a >= 0 ? a : l, s) + NN;
    const double* E1 = Fblk(F, d, b, l, s + 1) + NN;
    const double* B1 = Fblk(F, d, b, bb >= 0 ? bb : l, s + 1) + NN;
    // Every load is issued before the first LDS store, and the stores are unconditional on the
    // same clamped index (surplus lanes rewrite the last element with its own value): a store
    // under a lane predicate -- or a loop around load + store -- makes the loads complete one after
    // the other (six round trips per wavefront before this form).
    const double* zs = z + ((size_t)b * N + s) * ROWS;
    const double zx = zs[NX + (lane < W ? lane : W - 1)];
    const double zl = zs[ROWS + (lane < 2 * NX ? lane : 2 * NX - 1)];
    if constexpr (NX % 2 == 0) {  // 16-byte copies (block and row offsets are even)
      constexpr int N1 = W * NX / 2, I1 = (N1 + 63) / 64, N2 = NN / 2, I2 = (N2 + 63) / 64;
      double2 t0[I1], t1[I1], t2[I2], t3[I2];
#pragma unroll
      for (int it = 0; it < I1; ++it) {
        const int e = lane + 64 * it, ec = e < N1 ? e : N1 - 1;
        t0[it] = reinterpret_cast<const double2*>(Es)[ec];
        t1[it] = reinterpret_cast<const double2*>(Fas)[ec];
      }
#pragma unroll
      for (int it = 0; it < I2; ++it) {
        const int e = lane + 64 * it, ec = e < N2 ? e : N2 - 1;
        t2[it] = reinterpret_cast<const double2*>(E1)[ec];
        t3[it] = reinterpret_cast<const double2*>(B1)[ec];
      }
#pragma unroll
      for (int it = 0; it < I1; ++it) {
        const int e = lane + 64 * it, ec = e < N1 ? e : N1 - 1;
        reinterpret_cast<double2*>(in.Exu)[ec] = t0[it];
        reinterpret_cast<double2*>(in.Axu)[ec] = t1[it];
      }
#pragma unroll
      for (int it = 0; it < I2; ++it) {
        const int e = lane + 64 * it, ec = e < N2 ? e : N2 - 1;
        reinterpret_cast<double2*>(in.E1x)[ec] = t2[it];
        reinterpret_cast<double2*>(in.B1x)[ec] = t3[it];
      }
    } else {
      constexpr int N1 = W * NX, I1 = (N1 + 63) / 64, N2 = NN, I2 = (N2 + 63) / 64;
      double t0[I1], t1[I1], t2[I2], t3[I2];
#pragma unroll
      for (int it = 0; it < I1; ++it) {
        const int e = lane + 64 * it, ec = e < N1 ? e : N1 - 1;
        t0[it] = Es[ec]; t1[it] = Fas[ec];
      }
#pragma unroll
      for (int it = 0; it < I2; ++it) {
        const int e = lane + 64 * it, ec = e < N2 ? e : N2 - 1;
        t2[it] = E1[ec]; t3[it] = B1[ec];
      }
#pragma unroll
      for (int it = 0; it < I1; ++it) {
        const int e = lane + 64 * it, ec = e < N1 ? e : N1 - 1;
        in.Exu[ec] = t0[it]; in.Axu[ec] = t1[it];
      }
#pragma unroll
      for (int it = 0; it < I2; ++it) {
        const int e = lane + 64 * it, ec = e < N2 ? e : N2 - 1;
        in.E1x[ec] = t2[it]; in.B1x[ec] = t3[it];
      }
    }
    in.zxu[lane < W ? lane : W - 1] = zx;
    in.z1[lane < 2 * NX ? lane : 2 * NX - 1] = zl;
  }
  wave_lds_sync();
  SEG(0);
  double Lrow[NX];
  const bool bad = separator_core<NX, NU, STRICT, KEEP>(lane, ab, in, out, Lrow,
                                                       AB + ((size_t)b * N + s) * NX * W, W,
                                                       (!LAMBDA_OUT && store_l) ? Fblk(F, d, b, l, s + 1) : nullptr);
  if (bad && lane == 0) flag_failure(info, d, b);
  SEG(6);  // re-arms the clock after the core's own marks

  // stores: rows of the solved panel
  double* myrec = rec + ((size_t)b * N + s) * (2 * NN + NX);
  const int grp = lane / NX;
  if (grp < 2) {
    const int colidx = grp == 0 ? a : bb;
    if (colidx >= 0) {
      double row[NX];
#pragma unroll
      for (int c = 0; c < NX; ++c) row[c] = out.X[gi * LD + grp * NX + c];
      store_row<NX>(myrec + grp * NN + gi * NX, row);
      if constexpr (LAMBDA_OUT) store_row<NX>(Fblk(F, d, b, colidx, s + 1) + gi * NX, row);
    }
  } else if (grp == 2) {
    const double v = out.X[gi * LD + 2 * NX];
    myrec[2 * NN + gi] = v;
    z[((size_t)b * N + s + 1) * ROWS + gi] = v;
  }
  if constexpr (LAMBDA_OUT) { if (lane < NX) store_row<NX>(Fblk(F, d, b, l, s + 1) + gi * NX, Lrow); }
  SEG(4);
}

// leaf-phase rhs entry rr of knot i from the raw right-hand side (the rhs part of ndlqr_SolveLeaf)
template <int NX, int NU>
__device__ __forceinline__ double leaf_rhs_entry(const Dims& d, const int b, const int i, const int rr,
                                                 const double* __restrict__ QR, const double* __restrict__ rhs) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU;
  const double* r0 = rhs + ((size_t)b * d.N + i) * ROWS;
  const double* qr = QR + ((size_t)b * d.N + i) * W;
  const bool lam = rr < NX, last = (i == d.N - 1);
  if (i == 0) {
    if (lam) return fma(-qr[rr], r0[rr], -r0[NX + rr]);
    if (rr < 2 * NX) return -r0[rr - NX];
    return r0[rr] / qr[rr - NX];
  }
  if (lam) return r0[rr];
  if (rr < 2 * NX || !last) return r0[rr] / qr[rr - NX];
  return r0[rr];
}

// ------------------------------------------------------------------------------------- separator-only schedule
// Fast mode without KEEP, upper levels: instead of keeping the first and last knot of every
// subtree up to date (28-row states that the next separator re-reads), every eliminated
// separator s' pushes the Schur-complement contributions of the REDUCED system -- block cyclic
// reduction on the separators alone -- to its two neighbours A (left of its subtree) and B (right):
//     DR[A] += Y_a' Y_a     gR[A] += Y_a' y_z      DL[B] += Y_bb' Y_bb    gL[B] += Y_bb' y_z
//     coupling of the lower-level neighbour (the parent) to the other one:
//        left child  (parent B):  CA[B] = Y_bb' Y_a        right child (parent A):  CB[A] = Y_a' Y_bb
// with Y = L^-1 [r_a | r_bb | b~] the forward-substituted panel (f' r = Y' Y by symmetry). A
// separator s of a later level then starts from
//     S-bar = leafS_s - DL[s] - DR[s],   r_a = -CA[s],   r_bb = -CB[s],   b~ = leaf_s - gL[s] - gR[s]
// (leafS_s = A Q^-1 A' + B R^-1 B' + Q_{s+1}^-1, leaf_s as in rhs_forward_*): the same S-bar, f_a,
// f_bb, z_sep as the knot-based schedule up to rounding, from 12x12 blocks instead of knot rows.
// One slot per separator of level >= 2 (s = 3 mod 4): DL | DR | CA | CB | gL | gR.
template <int NX>
struct RedSlot {
  // USED doubles of a slot; slots are padded to whole 128-byte lines (SIZE), so that a wavefront
  // that reads its slot never pulls bytes of a neighbouring slot into its caches (tree schedule).
  // DL and DR are symmetric: their lower triangles alone, packed by rows (entry (r, c), c <= r, at r (r + 1) / 2 + c) --
  // the upper levels run at the rate the HBM delivers their slots and takes their pushes, and this is a fifth of it.
  static constexpr int NN = NX * NX, TRI = NX * (NX + 1) / 2, USED = 2 * TRI + 2 * NN + 2 * NX, SIZE = (USED + 15) / 16 * 16;
  double* p;
  __device__ __forceinline__ double* DL() const { return p; }
  __device__ __forceinline__ double* DR() const { return p + TRI; }
  __device__ __forceinline__ double* CA() const { return p + 2 * TRI; }
  __device__ __forceinline__ double* CB() const { return p + 2 * TRI + NN; }
  __device__ __forceinline__ double* gL() const { return p + 2 * TRI + 2 * NN; }
  __device__ __forceinline__ double* gR() const { return p + 2 * TRI + 2 * NN + NX; }
  // entry (r, c) of a packed symmetric block
  __host__ __device__ static constexpr int tri(const int r, const int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
};
template <int NX>
__device__ __forceinline__ RedSlot<NX> red_slot(double* red, const Dims& d, const int b, const int t) {
  return RedSlot<NX>{red + ((size_t)b * (d.N >> 2) + (t >> 2)) * RedSlot<NX>::SIZE};
}

// ------------------------------------------------------------------------------------- row update helpers
// One knot row through one level, in place (used by bottom_small and apply_small):
//   Ca <- (left ? Ca : 0) - E f_a,  Cb <- (left ? 0 : Cb) - E f_bb,  zz <- zz - E z_sep.
// A row that must not change this level (a lambda row that is not eliminated yet) takes part with
// E = 0: x + (-0 * f) == x bit for bit, so no lane needs a select around the FMAs and the created
// column of such a row comes out as the required zero. f(k, c) = f[k * LDF + c].
template <int NX, int LDF, bool STRICT>
__device__ __forceinline__ void row_update(double (&E)[NX], double (&Ca)[NX], double (&Cb)[NX], double& zz,
                                           const double* fa, const double* fb, const double* zsp,
                                           const int zstride, const bool has_a, const bool has_b,
                                           const bool left, const bool active) {
  // The column that gets created this level (Ca of a right-half knot, Cb of a left-half knot)
  // is zero on entry: the leaf phase, rotate_roles and the loads at lstart leave it so -- no
  // select needed. `left` only documents which one that is.
  (void)left;
#pragma unroll
  for (int c = 0; c < NX; ++c) E[c] = active ? E[c] : 0.0;
  if (has_a) {
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int c = 0; c < NX; ++c) Ca[c] = mad<STRICT>(-E[k], fa[k * LDF + c], Ca[c]);
  }
  if (has_b) {
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int c = 0; c < NX; ++c) Cb[c] = mad<STRICT>(-E[k], fb[k * LDF + c], Cb[c]);
  }
#pragma unroll
  for (int k = 0; k < NX; ++k) zz = mad<STRICT>(-E[k], zsp[k * zstride], zz);
}

// The same level step for a row that carries only its LIVE outer column C (column a for a
// left-half knot, bb for a right-half one) between levels: the other outer column is created
// here (D, from zero) -- one register array less across the separator than (E, Ca, Cb).
//   C <- C - E f_live,  D <- -E f_new,  zz <- zz - E z_sep,  then the roles of level l+1:
//   E <- (left == left child) ? D : C,  C <- the other one.
// f_live / f_new: f_a / f_bb for a left-half knot, swapped for a right-half one (the pointers may
// differ between the two knots of a wavefront: two LDS addresses per read, distinct banks).
template <int NX, int LDF, bool STRICT>
__device__ __forceinline__ void row_update_live(double (&E)[NX], double (&C)[NX], double& zz,
                                                const double* f_live, const double* f_new, const double* zsp,
                                                const bool has_live, const bool has_new, const bool active,
                                                const bool take_sep, const int r, const bool e_is_created) {
  double D[NX];
#pragma unroll
  for (int c = 0; c < NX; ++c) { E[c] = active ? E[c] : 0.0; D[c] = 0.0; }
  if (has_live) {
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int c = 0; c < NX; ++c) C[c] = mad<STRICT>(-E[k], f_live[k * LDF + c], C[c]);
  }
  if (has_new) {
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int c = 0; c < NX; ++c) D[c] = mad<STRICT>(-E[k], f_new[k * LDF + c], D[c]);
  }
#pragma unroll
  for (int k = 0; k < NX; ++k) zz = mad<STRICT>(-E[k], zsp[k * LDF], zz);
  if (take_sep) {  // lambda rows of knot s+1 receive the separator's results (its own panel rows)
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      if (has_live) C[c] = f_live[r * LDF + c];
      if (has_new) D[c] = f_new[r * LDF + c];
    }
    zz = zsp[r * LDF];
  }
#pragma unroll
  for (int c = 0; c < NX; ++c) {
    const double cn = C[c], dn = D[c];
    E[c] = e_is_created ? dn : cn;
    C[c] = e_is_created ? cn : dn;
  }
}

// Column roles of level l+1: a left child's right outer column is column l+1 (it becomes E),
// a right child's left outer column is. `left_child` must be wave-uniform (scalar branch).
template <int NX>
__device__ __forceinline__ void rotate_roles(double (&E)[NX], double (&Ca)[NX], double (&Cb)[NX],
                                             const bool left_child) {
  if (left_child) {
#pragma unroll
    for (int c = 0; c < NX; ++c) { E[c] = Cb[c]; Cb[c] = 0.0; }
  } else {
#pragma unroll
    for (int c = 0; c < NX; ++c) { E[c] = Ca[c]; Ca[c] = 0.0; }
  }
}

// ------------------------------------------------------------------------------------- Schur update
template <int NX, int NU>
struct SchurShape {
  static constexpr int ROWS = 2 * NX + NU;
  // knots per wavefront: two, i.e. exactly one level-0 subtree, so that the separator record is
  // wave-uniform at every level (scalar loads). Wider packing for ROWS <= 16 is left for later.
  static constexpr int KPW = 2;
  static_assert(2 * ROWS <= 64, "two knots (2 * (2 NX + NU) rows) must fit a wavefront");
  static constexpr int WAVES = 4;
  static constexpr int KPB = KPW * WAVES;    // knots per workgroup
  static constexpr int REC = 2 * NX * NX + NX;  // doubles per record: f_a | f_bb | z_sep
};

// Rows of knot i (row r = this lane) through level l: the Schur update of ndlqr_UpdateShurFactor
// restricted to the live columns, in two steps so that the loads can be issued long before the
// separator's results exist: schur_rows_load (operands into registers) and schur_rows_apply
// (arithmetic + stores). f(k, c) = fa[k * LDF + c], z_sep(k) = zsep[k * ZS].
template <int NX>
struct SchurRow {
  double E[NX];  // this row of column l
  double C[NX];  // this row of the knot's existing outer column (left half: a, right half: bb)
  double zz;     // its rhs entry
};

template <int NX, int NU>
__device__ __forceinline__ void schur_rows_load(const Dims& d, const int l, const int i, const int r, const int b,
                                                const double* F, const double* z, SchurRow<NX>& row) {
  constexpr int ROWS = 2 * NX + NU;
  const int N = d.N, half = 1 << l;
  const int base = (i >> (l + 1)) << (l + 1), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const bool left = i <= s;
  const bool calc_lambda = (i == 0) || (i & (half - 1)) != 0;
#pragma unroll
  for (int c = 0; c < NX; ++c) { row.E[c] = 0.0; row.C[c] = 0.0; }
  row.zz = 0.0;
  if (r < NX && !calc_lambda) return;
  load_row<NX>(Fblk(F, d, b, l, i) + r * NX, row.E);
  const int cc = left ? a : bb;
  if (cc >= 0) load_row<NX>(Fblk(F, d, b, cc, i) + r * NX, row.C);
  row.zz = z[((size_t)b * N + i) * ROWS + r];
}

template <int NX, int NU, bool STRICT, int LDF, int ZS>
__device__ __forceinline__ void schur_rows_apply(const Dims& d, const int l, const int i, const int r, const int b,
                                                 double* F, double* z, const double* fa, const double* fb,
                                                 const double* zsep, const SchurRow<NX>& row) {
  constexpr int ROWS = 2 * NX + NU;
  const int N = d.N, half = 1 << l;
  const int base = (i >> (l + 1)) << (l + 1), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const bool left = i <= s;
  const bool calc_lambda = (i == 0) || (i & (half - 1)) != 0;

  if (r < NX && !calc_lambda) {
    if (i != s + 1) {  // created blocks get explicit zero lambda rows (see schur_generic)
      if (a >= 0 && !left) {
        double* g = Fblk(F, d, b, a, i) + r * NX;
#pragma unroll
        for (int c = 0; c < NX; ++c) g[c] = 0.0;
      }
      if (bb >= 0 && left) {
        double* g = Fblk(F, d, b, bb, i) + r * NX;
#pragma unroll
        for (int c = 0; c < NX; ++c) g[c] = 0.0;
      }
    }
    return;
  }

  if (a >= 0) {
    double acc[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) acc[c] = left ? row.C[c] : 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int c = 0; c < NX; ++c) acc[c] = mad<STRICT>(-row.E[k], fa[k * LDF + c], acc[c]);
    store_row<NX>(Fblk(F, d, b, a, i) + r * NX, acc);
  }
  if (bb >= 0) {
    double acc[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) acc[c] = left ? 0.0 : row.C[c];
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int c = 0; c < NX; ++c) acc[c] = mad<STRICT>(-row.E[k], fb[k * LDF + c], acc[c]);
    store_row<NX>(Fblk(F, d, b, bb, i) + r * NX, acc);
  }
  {
    double accz = row.zz;
#pragma unroll
    for (int k = 0; k < NX; ++k) accz = mad<STRICT>(-row.E[k], zsep[k * ZS], accz);
    z[((size_t)b * N + i) * ROWS + r] = accz;
  }
}

template <int NX, int NU, bool STRICT, int LDF, int ZS>
__device__ __forceinline__ void schur_rows(const Dims& d, const int l, const int i, const int r, const int b,
                                           double* F, double* z, const double* fa, const double* fb,
                                           const double* zsep) {
  SchurRow<NX> row;
  schur_rows_load<NX, NU>(d, l, i, r, b, F, z, row);
  schur_rows_apply<NX, NU, STRICT, LDF, ZS>(d, l, i, r, b, F, z, fa, fb, zsep, row);
}

// ------------------------------------------------------------------------------------- upper levels
// One level of the knot-based schedule, one wavefront per subtree: the separator (separator_wave)
// and, below the top level, the Schur update of the subtree's first and last knot -- the two rows
// later separators read -- straight from the solved panel in LDS.  grid (N >> (l+1), batch), block 64.
template <int NX, int NU, bool STRICT, bool KEEP>
__global__ __launch_bounds__(64) void level_small(Dims d, int l, const double* __restrict__ AB, double* F,
                                                  double* z, double* __restrict__ rec, int* __restrict__ info,
                                                  const int store_l) {
  constexpr int ROWS = 2 * NX + NU, LD = SepOut<NX>::LD;
  __shared__ SepIn<NX, NU> in;
  __shared__ SepOut<NX> out;
  const int lane = threadIdx.x, sub = blockIdx.x, b = blockIdx.y;
  separator_wave<NX, NU, STRICT, KEEP, KEEP>(d, l, sub, b, lane, AB, F, z, rec, info, in, out, store_l != 0);
  SEG_INIT();
  const int kn = lane / ROWS, r = lane - kn * ROWS;
  if (l < d.K - 1 && kn < 2) {
    const int T = 2 << l;
    const int i = sub * T + (kn == 0 ? 0 : T - 1);
    schur_rows<NX, NU, STRICT, LD, LD>(d, l, i, r, b, F, z, out.X, out.X + NX, out.X + 2 * NX);
  }
#ifdef NDLQR_SEGTIME
  __builtin_amdgcn_s_waitcnt(0);
#endif
  SEG(5);
}

// ------------------------------------------------------------------------------------- apply
// All upper levels J..K-1 for every knot in ONE pass (DESIGN.md "boundary-first"): once the
// separator records of those levels exist (level_small: separator + the boundary knots of every
// subtree), a knot's updates at successive levels only involve its
// own rows: E (column l), the two live outer columns and its rhs entry stay in registers and
// rotate from level to level; only the rhs (and, with KEEP, the finished columns) go back to HBM.
// Same operations in the same order per element as a full Schur pass per level.
//   grid (N / KPB, batch), block 256, dynamic LDS = (K - J) * REC doubles.
// Knots that the boundary pass already advanced (first / last knot of a 2^J block) join at the
// level where that pass left them (lstart).
template <int NX, int NU, bool STRICT, bool KEEP>
__global__ __launch_bounds__(256) void apply_small(Dims d, int J, double* F, double* z,
                                                   const double* __restrict__ recs) {
  using Sh = SchurShape<NX, NU>;
  constexpr int ROWS = Sh::ROWS, KPW = Sh::KPW, KPB = Sh::KPB, REC = Sh::REC;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int N = d.N, K = d.K, b = blockIdx.y;
  const int first = blockIdx.x * KPB;
  {
    // records of the levels J..K-1 into LDS: every load before the first store (a loop around
    // load + store completes them one after the other); levels beyond K-1 / elements beyond REC are
    // clamped, the surplus stores rewrite an element with its own value
    constexpr int MAXL = 12, IT = (REC + 255) / 256;
    const int nl = K - J;
    if (nl <= MAXL) {
      double t[MAXL][IT];
#pragma unroll
      for (int q = 0; q < MAXL; ++q) {
        const int l = J + (q < nl ? q : nl - 1);
        const int qs = ((first >> (l + 1)) << (l + 1)) + (1 << l) - 1;
        const double* src = recs + ((size_t)b * N + qs) * REC;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int e = threadIdx.x + 256 * it;
          t[q][it] = src[e < REC ? e : REC - 1];
        }
      }
#pragma unroll
      for (int q = 0; q < MAXL; ++q) {
        double* dst = lds + (q < nl ? q : nl - 1) * REC;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int e = threadIdx.x + 256 * it;
          dst[e < REC ? e : REC - 1] = t[q][it];
        }
      }
    } else {
      for (int l = J; l < K; ++l) {
        const int qs = ((first >> (l + 1)) << (l + 1)) + (1 << l) - 1;
        const double* src = recs + ((size_t)b * N + qs) * REC;
        double* dst = lds + (l - J) * REC;
        for (int e = threadIdx.x; e < REC; e += 256) dst[e] = src[e];
      }
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kn = lane / ROWS, r = lane - kn * ROWS;
  if (kn >= KPW) return;
  const int i = first + wave * KPW + kn;
  const bool lam = r < NX;

  int lstart = J;
  {
    const int mask = (1 << J) - 1;
    if ((i & mask) == 0) lstart = (i == 0) ? K : __builtin_ctz(i);
    else if ((i & mask) == mask) lstart = trailing_ones(i);
    if (lstart > K - 1) lstart = K - 1;
  }

  double E[NX], Ca[NX], Cb[NX];
  double zz = 0.0;
#pragma unroll
  for (int c = 0; c < NX; ++c) { E[c] = 0.0; Ca[c] = 0.0; Cb[c] = 0.0; }
  double* zp = z + ((size_t)b * N + i) * ROWS + r;

  for (int l = J; l < K; ++l) {
    if (l < lstart) continue;
    const int half = 1 << l, T = 2 << l;
    const int base = (i >> (l + 1)) << (l + 1), s = base + half - 1;
    int a, bb;
    outer_columns(base, l, N, a, bb);
    const bool left = i <= s;
    const bool calc_lambda = (i == 0) || (i & (half - 1)) != 0;
    const bool active = !lam || calc_lambda;
    const double* rc = lds + (l - J) * REC;
    const double* fa = rc;
    const double* fb = rc + NX * NX;
    const double* zsep = rc + 2 * NX * NX;

    if (l == lstart) {  // pick the knot up where the level-by-level kernels left it
      load_row<NX>(Fblk(F, d, b, l, i) + r * NX, E);
      if (left) { if (a >= 0) load_row<NX>(Fblk(F, d, b, a, i) + r * NX, Ca); }
      else      { if (bb >= 0) load_row<NX>(Fblk(F, d, b, bb, i) + r * NX, Cb); }
      zz = *zp;
    } else if (KEEP && active) {
      store_row<NX>(Fblk(F, d, b, l, i) + r * NX, E);  // column l is final for this knot
    }

    // for l >= J >= 1 both knots of a wavefront sit in the same half of the same subtree
    const bool left_u = __builtin_amdgcn_readfirstlane((int)left) != 0;
    row_update<NX, NX, STRICT>(E, Ca, Cb, zz, fa, fb, zsep, 1, a >= 0, bb >= 0, left_u, active);
    if (!active && i == s + 1) {  // lambda rows of knot s+1 receive the separator's results
#pragma unroll
      for (int c = 0; c < NX; ++c) {
        if (a >= 0) Ca[c] = fa[r * NX + c];
        if (bb >= 0) Cb[c] = fb[r * NX + c];
      }
      zz = zsep[r];
    }
    rotate_roles<NX>(E, Ca, Cb, __builtin_amdgcn_readfirstlane((int)((base & T) == 0)) != 0);
  }
  *zp = zz;
}

// ------------------------------------------------------------------------------------- back-substitution
// Fast mode without KEEP: the solution straight from the separator records and the problem data
// (the classic nested-dissection back-substitution; replaces hand-off + finish_small).
//   multipliers, top-down over the tree:  y_s = z_sep(s) - f_a(s) y_A - f_bb(s) y_B,
//       A / B = the separators left / right of s's subtree (columns a / bb), absent at the ends;
//   lambda_k = y_{k-1}; knot 0: lambda = Q x0 + q + A_0' y_0;
//   x_k = Q^-1 (-q_k - A_k' y_k + y_{k-1}),  u_k = R^-1 (-r_k - B_k' y_k)     (stationarity rows)
// -- the same quantities the level-by-level sweep produces, up to rounding. A workgroup owns 8
// consecutive knots: the 7 separators inside them and the K - 3 on their path to the root; every
// (separator, row) pair is fetched by one thread up front, then the levels resolve through LDS.
//   grid (N / 8, batch), block 256; requires N >= 8 and (K + 4) * NX <= 256.
template <int NX, int NU>
__global__ __launch_bounds__(256) void backsub_small(Dims d, const double* __restrict__ AB,
                                                     const double* __restrict__ QR,
                                                     const double* __restrict__ rhs,
                                                     const double* __restrict__ recs, double* __restrict__ z) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, REC = 2 * NN + NX, KPB = 8, MAXSLOT = 256 / NX;
  __shared__ double ys[MAXSLOT][NX];
  const int N = d.N, K = d.K, b = blockIdx.y, first = (blockIdx.x + d.xoff) * KPB;  // (xoff: a knot range alone, NDLQR_SOLN_ONLY)
  const int npath = K - 3, nslot = npath + 7;
  // slot -> separator: slots 0..npath-1 the path (level K-1 first), then the 7 local ones
  auto slot_sep = [&](int q, int& s, int& l) {
    if (q < npath) { l = K - 1 - q; s = ((first >> (l + 1)) << (l + 1)) + (1 << l) - 1; }
    else { s = first + (q - npath); l = trailing_ones(s); }
  };
  auto sep_slot = [&](int s) -> int {  // only called for separators of this workgroup's scope
    return (s >= first && s < first + 7) ? npath + (s - first) : K - 1 - trailing_ones(s);
  };

  // ---- every thread's operands, requested before anything waits
  const int t = threadIdx.x;
  const int q = t / NX, r = t - q * NX;
  int s = 0, l = 0, slotA = -1, slotB = -1;
  double fa[NX], fb[NX], zs = 0.0;
  const bool sep_thread = q < nslot;
  if (sep_thread) {
    slot_sep(q, s, l);
    const int base = s - ((1 << l) - 1);
    const bool hasA = base > 0, hasB = base + (2 << l) < N;
    const double* rc = recs + ((size_t)b * N + s) * REC;
    zs = rc[2 * NN + r];
    if (hasA) { load_row<NX>(rc + r * NX, fa); slotA = sep_slot(base - 1); }
    if (hasB) { load_row<NX>(rc + NN + r * NX, fb); slotB = sep_slot(base + (2 << l) - 1); }
  }
  const int kn = t / ROWS, rr = t - kn * ROWS;  // output role: knot kn of the workgroup, row rr
  const bool out_thread = kn < KPB;
  const int i = first + (out_thread ? kn : 0);
  const bool lam = rr < NX;
  const int col = lam ? rr : rr - NX;  // column of [A_i | B_i] this row dots with y_i
  const bool needs_ab = out_thread && i < N - 1 && (lam ? i == 0 : !(i == 0 && rr < 2 * NX));
  double abcol[NX], rv = 0.0, rv2 = 0.0, qv = 1.0;
  if (out_thread) {
    const double* r0 = rhs + ((size_t)b * N + i) * ROWS;
    const double* qr = QR + ((size_t)b * N + i) * W;
    rv = r0[rr];
    if (i == 0 && lam) { rv2 = r0[NX + rr]; qv = qr[rr]; }
    else if (!lam) qv = qr[rr - NX];
    if (needs_ab) {
      const double* abk = AB + ((size_t)b * N + i) * NX * W + col;
#pragma unroll
      for (int c = 0; c < NX; ++c) abcol[c] = abk[c * W];
    }
  }

  // ---- multipliers, one tree level per step
  for (int L = K - 1; L >= 0; --L) {
    if (sep_thread && l == L) {
      double acc = zs;
      if (slotA >= 0) {
#pragma unroll
        for (int c = 0; c < NX; ++c) acc = fma(-fa[c], ys[slotA][c], acc);
      }
      if (slotB >= 0) {
#pragma unroll
        for (int c = 0; c < NX; ++c) acc = fma(-fb[c], ys[slotB][c], acc);
      }
      ys[q][r] = acc;
    }
    __syncthreads();
  }

  // ---- solution rows
  if (!out_thread) return;
  double out;
  const double* yi = ys[sep_slot(i < N - 1 ? i : i - 1)];        // y_i   (unused for the last knot)
  const double* yp = ys[sep_slot(i > 0 ? i - 1 : 0)];            // y_{i-1} (unused for knot 0)
  double dot = 0.0;
  if (needs_ab) {
#pragma unroll
    for (int c = 0; c < NX; ++c) dot = fma(abcol[c], yi[c], dot);
  }
  if (lam) {
    out = (i == 0) ? fma(-qv, rv, -rv2) + dot : yp[rr];
  } else if (rr < 2 * NX) {
    out = (i == 0) ? -rhs[((size_t)b * N) * ROWS + (rr - NX)] : (rv - dot + yp[rr - NX]) / qv;
  } else {
    out = (i == N - 1) ? rv : (rv - dot) / qv;
  }
  z[((size_t)b * N + i) * ROWS + rr] = out;
}

// ------------------------------------------------------------------------------------- rhs-only re-solve
// New right-hand side against a cached factorisation (fast mode, NDLQR_FLAG_KEEP_FACT): the
// forward half of the nested-dissection solve on the separators alone, then backsub_small.
//   b~_s   = leaf_s - sum_{l' < l} [ f_bb(s - 2^l')' b~_{s - 2^l'} + f_a(s + 2^l')' b~_{s + 2^l'} ]
//   z_sep(s) = (L_s L_s')^-1 b~_s
// leaf_s = A_s z.x + B_s z.u - z(s+1).x - z(s+1).lambda on the leaf-phase rhs (what the inner
// product of level 0 forms); s -/+ 2^l' are the separators of the two child chains whose subtrees
// end / start at s -- by symmetry of the reduced system their coupling to s is the transpose
// of their own f_bb / f_a. Reads per separator: its record and factor, nothing of the knots'
// factor columns (the level-by-level sweep streams all N K of them).
// (L L')^-1 applied to one vector by one wavefront: lane r < NX holds entry r of the vector, row
// r of L (Lrow[j] = L(r, j), j <= r) and column r of L (Lcol[j] = L(j, r), j >= r).
template <int NX>
__device__ __forceinline__ double chol_solve_wave(double v, const double (&Lrow)[NX], const double (&Lcol)[NX],
                                                  const int lane) {
  double dinv = 1.0;  // lane j: 1 / L(j, j)
#pragma unroll
  for (int j = 0; j < NX; ++j) dinv = (lane == j) ? 1.0 / Lrow[j] : dinv;
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const double xj = readlane_f64(v, j) * readlane_f64(dinv, j);
    v = (lane == j) ? xj : ((lane > j) ? fma(-Lrow[j], xj, v) : v);
  }
#pragma unroll
  for (int j = NX - 1; j >= 0; --j) {
    const double xj = readlane_f64(v, j) * readlane_f64(dinv, j);
    v = (lane == j) ? xj : ((lane < j) ? fma(-Lcol[j], xj, v) : v);
  }
  return v;
}

// Operands of one separator's forward step, fetched by its wavefront in one round trip:
// row / column rc of the factor and, per child-chain level, column rc of f_bb(s - 2^lp) and of
// f_a(s + 2^lp) (the transposed products read columns).
template <int NX, int NLP>
struct ForwardOps {
  double Lrow[NX], Lcol[NX];
  double fl[NLP > 0 ? NLP : 1][NX], fr[NLP > 0 ? NLP : 1][NX];
};

template <int NX, int NU, int NLP>
__device__ __forceinline__ void forward_fetch(const Dims& d, const int b, const int s, const int l, const int lp0,
                                              const int rc, const double* F, const double* recs,
                                              ForwardOps<NX, NLP>& op) {
  constexpr int NN = NX * NX, REC = 2 * NN + NX;
  const double* Lb = Fblk(F, d, b, l, s + 1);  // lambda rows of knot s+1, column l: the factor
  load_row<NX>(Lb + rc * NX, op.Lrow);
#pragma unroll
  for (int j = 0; j < NX; ++j) op.Lcol[j] = Lb[j * NX + rc];
#pragma unroll
  for (int q = 0; q < NLP; ++q) {
    const int lp = lp0 + q;
    if (lp < l) {
      const double* fbb = recs + ((size_t)b * d.N + (s - (1 << lp))) * REC + NN;
      const double* fa = recs + ((size_t)b * d.N + (s + (1 << lp))) * REC;
#pragma unroll
      for (int c = 0; c < NX; ++c) { op.fl[q][c] = fbb[c * NX + rc]; op.fr[q][c] = fa[c * NX + rc]; }
    }
  }
}

// b~ = acc - child-chain terms (bt_of(s') = b~ of separator s'), z_sep = (L L')^-1 b~ -> record
template <int NX, int NU, int NLP, class BtOf>
__device__ __forceinline__ double forward_finish(const Dims& d, const int b, const int s, const int l,
                                                 const int lp0, double acc, const int lane, double* recs,
                                                 const ForwardOps<NX, NLP>& op, BtOf bt_of) {
  constexpr int NN = NX * NX, REC = 2 * NN + NX;
#pragma unroll
  for (int q = 0; q < NLP; ++q) {
    const int lp = lp0 + q;
    if (lp < l) {
      const double* bl = bt_of(s - (1 << lp));
      const double* br = bt_of(s + (1 << lp));
#pragma unroll
      for (int c = 0; c < NX; ++c) acc = fma(-op.fl[q][c], bl[c], acc);
#pragma unroll
      for (int c = 0; c < NX; ++c) acc = fma(-op.fr[q][c], br[c], acc);
    }
  }
  const double zs = chol_solve_wave<NX>(acc, op.Lrow, op.Lcol, lane);
  if (lane < NX) recs[((size_t)b * d.N + s) * REC + 2 * NN + lane] = zs;
  return acc;
}

// (L L')^-1 applied to four vectors at once, one per 16-lane group of a wavefront (lane r of a
// group: entry r, row r and column r of that group's factor); broadcasts are width-16 shuffles.
template <int NX>
__device__ __forceinline__ double chol_solve_group16(double v, const double (&Lrow)[NX], const double (&Lcol)[NX],
                                                     const int r) {
  double diag = 1.0;
#pragma unroll
  for (int j = 0; j < NX; ++j) diag = (r == j) ? Lrow[j] : diag;
  const double dinv = 1.0 / diag;
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const double xj = __shfl(v * dinv, j, 16);
    v = (r == j) ? xj : ((r > j) ? fma(-Lrow[j], xj, v) : v);
  }
#pragma unroll
  for (int j = NX - 1; j >= 0; --j) {
    const double xj = __shfl(v * dinv, j, 16);
    v = (r == j) ? xj : ((r < j) ? fma(-Lcol[j], xj, v) : v);
  }
  return v;
}

// Levels 0..2 inside every block of 8 knots by ONE wavefront: 16-lane group g takes separator
// 2g (level 0), 4g+1 (level 1, g < 2), 3 (level 2, g = 0); idle groups shadow a live one and
// store nothing. Small workgroups keep thousands of them in flight, which is what hides the
// three dependent memory round trips. Also writes the block's share of the sums of the upper
// separators (left at z(first).x, right at z(first+1).x -- z is rewritten by backsub_small).
//   grid (N / 8, batch), block 64; NX <= 16.
template <int NX, int NU>
__global__ __launch_bounds__(64) void rhs_forward_small(Dims d, const double* __restrict__ AB,
                                                        const double* __restrict__ QR,
                                                        const double* __restrict__ rhs, const double* F,
                                                        double* recs, double* z) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU, NN = NX * NX, REC = 2 * NN + NX;
  static_assert(NX <= 16, "one separator row per lane of a 16-lane group");
  __shared__ double zl[8][ROWS];
  __shared__ double bt[7][NX];
  const int N = d.N, b = blockIdx.y, first = blockIdx.x * 8;
  const int lane = threadIdx.x, g = lane >> 4, r = lane & 15;
  const int rc = r < NX ? r : NX - 1;
  for (int e = lane; e < 8 * ROWS; e += 64) {
    const int kn = e / ROWS, rr = e - kn * ROWS;
    zl[kn][rr] = leaf_rhs_entry<NX, NU>(d, b, first + kn, rr, QR, rhs);
  }
  wave_lds_sync();
#pragma unroll
  for (int lvl = 0; lvl < 3; ++lvl) {
    const bool act = g < (4 >> lvl) && r < NX;
    const int gg = g & ((4 >> lvl) - 1);
    const int j = (gg << (lvl + 1)) + (1 << lvl) - 1, s = first + j;
    const double* Lb = Fblk(F, d, b, lvl, s + 1);  // lambda rows of knot s+1, column lvl: the factor
    double Lrow[NX], Lcol[NX], abrow[W];
    load_row<NX>(Lb + rc * NX, Lrow);
#pragma unroll
    for (int q = 0; q < NX; ++q) Lcol[q] = Lb[q * NX + rc];
    load_row<W>(AB + (((size_t)b * N + s) * NX + rc) * W, abrow);
    double acc = -zl[j + 1][rc];
#pragma unroll
    for (int k = 0; k < W; ++k) acc = fma(abrow[k], zl[j][NX + k], acc);
    acc -= zl[j + 1][NX + rc];
#pragma unroll
    for (int lp = 0; lp < lvl; ++lp) {
      const int jl = j - (1 << lp), jr = j + (1 << lp);
      const double* fbb = recs + ((size_t)b * N + first + jl) * REC + NN;
      const double* fa = recs + ((size_t)b * N + first + jr) * REC;
#pragma unroll
      for (int c = 0; c < NX; ++c) acc = fma(-fbb[c * NX + rc], bt[jl][c], acc);
#pragma unroll
      for (int c = 0; c < NX; ++c) acc = fma(-fa[c * NX + rc], bt[jr][c], acc);
    }
    const double zs = chol_solve_group16<NX>(acc, Lrow, Lcol, r);
    if (act) {
      bt[j][r] = acc;
      recs[((size_t)b * N + s) * REC + 2 * NN + r] = zs;
    }
    wave_lds_sync();
  }
  if (g < 2 && r < NX) {
    // group 0: separators whose subtree starts at `first` -> terms of separator first-1;
    // group 1: subtrees ending at first+7 -> terms of separator first+7
    const bool left = g == 0;
    if (left ? first > 0 : first + 8 < N) {
      double a2 = 0.0;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int jq = left ? (q == 0 ? 0 : (q == 1 ? 1 : 3)) : (q == 0 ? 6 : (q == 1 ? 5 : 3));
        const double* f = recs + ((size_t)b * N + first + jq) * REC + (left ? 0 : NN);
#pragma unroll
        for (int c = 0; c < NX; ++c) a2 = fma(f[c * NX + r], bt[jq][c], a2);
      }
      z[((size_t)b * N + first + (left ? 0 : 1)) * ROWS + NX + r] = a2;
    }
  }
}

// Levels 3..K-1 of one problem: one wavefront per separator, level by level.
//   grid (batch), block 512, dynamic LDS = (N / 8) * NX doubles.
template <int NX, int NU>
__global__ __launch_bounds__(512) void rhs_forward_upper(Dims d, const double* __restrict__ AB,
                                                         const double* __restrict__ QR,
                                                         const double* __restrict__ rhs, const double* F,
                                                         double* recs, const double* z) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU;
  extern __shared__ double btu[];  // b~ of separator 8 m + 7 at btu[m * NX]
  const int N = d.N, K = d.K, b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int rc = lane < NX ? lane : NX - 1;
  for (int lvl = 3; lvl < K; ++lvl) {
    const int nsep = N >> (lvl + 1);
    for (int q = wave; q < nsep; q += nw) {
      const int s = q * (2 << lvl) + (1 << lvl) - 1;
      const double zxu = lane < W ? leaf_rhs_entry<NX, NU>(d, b, s, NX + lane, QR, rhs) : 0.0;
      const double* abrow = AB + (((size_t)b * N + s) * NX + rc) * W;
      double acc = -leaf_rhs_entry<NX, NU>(d, b, s + 1, rc, QR, rhs);
#pragma unroll
      for (int k = 0; k < W; ++k) acc = fma(abrow[k], readlane_f64(zxu, k), acc);
      acc -= leaf_rhs_entry<NX, NU>(d, b, s + 1, NX + rc, QR, rhs);
      // levels 0..2 of the two neighbouring blocks
      acc -= z[((size_t)b * N + (s - 7) + 1) * ROWS + NX + rc];
      acc -= z[((size_t)b * N + (s + 1)) * ROWS + NX + rc];
      // child chains of levels 3..lvl-1, two levels per fetch; the last fetch carries the factor
      auto bt_of = [&](int sp) -> const double* { return btu + ((sp - 7) >> 3) * NX; };
      int lp0 = 3;
      for (; lp0 + 2 < lvl; lp0 += 2) {
        ForwardOps<NX, 2> op;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int lp = lp0 + q;
          const double* fbb = recs + ((size_t)b * N + (s - (1 << lp))) * (2 * NX * NX + NX) + NX * NX;
          const double* fa = recs + ((size_t)b * N + (s + (1 << lp))) * (2 * NX * NX + NX);
          const double* bl = bt_of(s - (1 << lp));
          const double* br = bt_of(s + (1 << lp));
#pragma unroll
          for (int c = 0; c < NX; ++c) { op.fl[q][c] = fbb[c * NX + rc]; op.fr[q][c] = fa[c * NX + rc]; }
#pragma unroll
          for (int c = 0; c < NX; ++c) acc = fma(-op.fl[q][c], bl[c], acc);
#pragma unroll
          for (int c = 0; c < NX; ++c) acc = fma(-op.fr[q][c], br[c], acc);
        }
      }
      ForwardOps<NX, 2> op;
      forward_fetch<NX, NU, 2>(d, b, s, lvl, lp0, rc, F, recs, op);
      const double btl = forward_finish<NX, NU, 2>(d, b, s, lvl, lp0, acc, lane, recs, op, bt_of);
      if (lane < NX) btu[((s - 7) >> 3) * NX + lane] = btl;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------- bottom
// Leaf phase + tree levels 0..JB-1 in ONE launch, everything on chip: a workgroup owns 2^JB
// consecutive knots, each wavefront two of them (lane = (knot, row)) with its rows of E, of the
// two outer columns and its rhs entry in registers from the leaf computation to the hand-off.
// Per level: the knots next to a separator publish the rows the separator needs in LDS; every
// wavefront computes the separator of its own subtree (redundantly for l > 0 -- no result
// broadcast, no idle waves), updates its two knots and rotates the column roles (see
// apply_small). Written back: column JB and the live outer column of every knot plus its rhs
// block -- what level_small(JB) / apply_small expect -- and, with KEEP, the finished columns
// 0..JB-1. Arithmetic and order per element are those of leaf_generic + separator_generic +
// schur_generic run level by level.
//   grid (N >> JB, batch), block 32 << JB threads. Requires N > 2^JB.
template <int NX, int NU, bool STRICT, bool KEEP, int JB>
__global__ __launch_bounds__(32 << JB, (KEEP && !STRICT) ? 2 : 3) void bottom_small(Dims d, const double* __restrict__ AB,
                                                         const double* __restrict__ QR,
                                                         const double* __restrict__ rhs, double* F,
                                                         double* z, int* __restrict__ info,
                                                         double* __restrict__ rec, const int lean,
                                                         const int recout) {
  // lean (fast mode without KEEP only): the solution comes from backsub_small, which needs the
  // records of the on-chip separators but nothing of the interior knots -- hand off only the
  // first and the last knot of the workgroup (what the upper levels read).
  // recout bit 0 (fast mode): write the records of the on-chip separators (lean, or KEEP for the
  // record-based right-hand-side re-solve); bit 1: also their Cholesky factors (KEEP_RECORDS).
  constexpr int W = NX + NU, ROWS = 2 * NX + NU;
  constexpr int NK = 1 << JB, NWAVE = NK / 2;
  static_assert(2 * ROWS <= 64 && 3 * NX <= 64, "two knots per wavefront, three lane groups of NX");
  // row pitch of the staged [A | B]: even W padded by two doubles so that the 12-16 row reads of
  // the separator (one row per lane, 16-byte LDS reads) fall into distinct banks
  constexpr int WP = (W % 2 == 0) ? W + 2 : W;
  struct alignas(16) Priv {   // per wavefront
    double ab[2][NX * WP];    // [A | B] of the wavefront's two knots (leaf phase, separators)
  };
  __shared__ SepIn<NX, NU> xs[NK / 2];   // per subtree of the current level: separator operands
  __shared__ SepOut<NX> so[NK / 2];      // per subtree: solved right-hand sides
  __shared__ Priv pv[NWAVE];
  constexpr int LD = SepOut<NX>::LD;

  const int N = d.N, b = blockIdx.y;
  const int wgbase = blockIdx.x * NK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kn = lane / ROWS, r = lane - kn * ROWS;
  const bool has_knot = kn < 2;
  const int i = wgbase + 2 * wave + (has_knot ? kn : 1);  // idle lanes shadow knot 1 (no stores)
  const bool lam = r < NX;
  Priv& me = pv[wave];

  SEG_INIT();
  // ---- stage [A | B] of my two knots (contiguous in memory); the leaf's own operands are
  //      requested in the same round trip
  const double* qr = QR + ((size_t)b * N + i) * W;
  const double* r0 = rhs + ((size_t)b * N + i) * ROWS;
  const double qv_in = qr[lam ? r : r - NX];
  const double rv = r0[r];
  {
    const double* src = AB + ((size_t)b * N + wgbase + 2 * wave) * NX * W;
    // all loads before the first LDS store, stores unconditional on the clamped index (a loop
    // around load + store completes the loads one after the other)
    if constexpr (W % 2 == 0) {
      // two knots = 2 NX rows of W doubles; the pair starts 16-byte aligned
      constexpr int NA = NX * W, IA = (NA + 63) / 64;
      double2 t[IA];
#pragma unroll
      for (int it = 0; it < IA; ++it) {
        const int e = lane + 64 * it;
        t[it] = reinterpret_cast<const double2*>(src)[e < NA ? e : NA - 1];
      }
#pragma unroll
      for (int it = 0; it < IA; ++it) {
        const int e = lane + 64 * it, ec = e < NA ? e : NA - 1;
        const int row = ec / (W / 2), c2 = ec - row * (W / 2);
        const int knot = row / NX, rr = row - knot * NX;
        reinterpret_cast<double2*>(&me.ab[knot][rr * WP])[c2] = t[it];
      }
    } else {
      constexpr int NA = 2 * NX * W, IA = (NA + 63) / 64;
      double t[IA];
#pragma unroll
      for (int it = 0; it < IA; ++it) { const int e = lane + 64 * it; t[it] = src[e < NA ? e : NA - 1]; }
#pragma unroll
      for (int it = 0; it < IA; ++it) {
        const int e = lane + 64 * it, ec = e < NA ? e : NA - 1, knot = ec / (NX * W);
        me.ab[knot][ec - knot * NX * W] = t[it];
      }
    }
  }
  __syncthreads();
  SEG(20);

  // ---- leaf phase in registers (ndlqr_SolveLeaf): own block O, block P towards the previous knot.
  //      State of a row between levels: E (column l) and C, its LIVE outer column -- column a for
  //      a knot in the left half of the next subtree, column bb for one in the right half.
  double E[NX], C[NX], zz;
  {
    const double* abk = me.ab[has_knot ? kn : 1];
    const bool last = (i == N - 1);
    // Row scaling by the diagonal Q (state rows) / R (input rows). STRICT reproduces the
    // reference's two divisions by L = q / sqrt(q) of its dense Cholesky solve; the fast mode
    // multiplies by one reciprocal (L * L == q up to rounding).
    double sc = 1.0, rq = 1.0;
    if (!lam) {
      const double qv = qv_in;
      if constexpr (STRICT) sc = qv / sqrt(qv); else rq = 1.0 / qv;
      if (has_knot && !(qv > 0.0) && !(last && r >= 2 * NX)) flag_failure(info, d, b);
    }
    auto scale = [&](double v) -> double {
      if constexpr (STRICT) return (v / sc) / sc; else return v * rq;
    };
    double O[NX], P[NX];
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      double o;
      if (lam) o = (i == 0) ? -abk[c * WP + r] : 0.0;
      else if (i == 0 && r < 2 * NX) o = 0.0;
      else o = scale(abk[c * WP + (r - NX)]);
      O[c] = last ? 0.0 : o;
      P[c] = (!lam && r < 2 * NX && c == r - NX) ? scale(-1.0) : 0.0;
    }
    const bool even = (i & 1) == 0;
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      E[c] = even ? O[c] : P[c];
      C[c] = even ? (i > 0 ? P[c] : 0.0) : O[c];  // even knot: left half at level 0 (column a), odd: right (bb)
    }
    if (i == 0) {
      if (lam) zz = mad<STRICT>(-qv_in, rv, -r0[NX + r]);
      else if (r < 2 * NX) zz = -r0[r - NX];
      else zz = scale(rv);
    } else {
      if (lam) zz = rv;
      else if (r < 2 * NX || !last) zz = scale(rv);
      else zz = rv;
    }
  }

  SEG(21);
  // ---- levels 0 .. JB-1
#pragma unroll
  for (int l = 0; l < JB; ++l) {
    const int half = 1 << l, T = 2 << l;
    // both knots of a wavefront sit in the same subtree at every level: tree quantities are
    // wave-uniform (scalar registers, scalar branches)
    const int i0 = __builtin_amdgcn_readfirstlane(wgbase + 2 * wave);
    const int base = (i0 >> (l + 1)) << (l + 1), s = base + half - 1;
    int a, bb;
    outer_columns(base, l, N, a, bb);
    const int sub = (base - wgbase) >> (l + 1);
    SepIn<NX, NU>& xc = xs[sub];
    SepOut<NX>& sout = so[sub];
    const bool left = (l == 0) ? (i == i0) : (i0 <= s);
    const bool calc_lambda = (i == 0) || (i & (half - 1)) != 0;
    const bool active = !lam || calc_lambda;
    const bool owner = (s + 1 - wgbase) / 2 == wave;  // the wavefront that holds knot s+1

    // publish what the separator needs from knots s and s+1
    if (has_knot && i == s && !lam) {
#pragma unroll
      for (int c = 0; c < NX; ++c) { xc.Exu[(r - NX) * NX + c] = E[c]; xc.Axu[(r - NX) * NX + c] = C[c]; }
      xc.zxu[r - NX] = zz;
    }
    if (has_knot && i == s + 1) {
      if (lam) xc.z1[r] = zz;
      else if (r < 2 * NX) {
#pragma unroll
        for (int c = 0; c < NX; ++c) { xc.E1x[(r - NX) * NX + c] = E[c]; xc.B1x[(r - NX) * NX + c] = C[c]; }
        xc.z1[NX + (r - NX)] = zz;
      }
    }
    __syncthreads();
    SEG(22);

    // separator of the subtree: computed once, by the wavefront that holds knot s+1; the other
    // wavefronts of the subtree wait at the barrier (their issue slots go to other workgroups)
    if (owner) {
      double Lrow[NX], ab[W];
      const int gi = lane % NX;
      {  // row gi of [A_s | B_s] from the staged copy of the wavefront that holds knot s
        const double* abs_ = pv[(s - wgbase) >> 1].ab[(s - wgbase) & 1] + gi * WP;
        if constexpr (WP % 2 == 0) {
#pragma unroll
          for (int k = 0; k < W / 2; ++k) {
            const double2 t = reinterpret_cast<const double2*>(abs_)[k];
            ab[2 * k] = t.x; ab[2 * k + 1] = t.y;
          }
        } else {
#pragma unroll
          for (int k = 0; k < W; ++k) ab[k] = abs_[k];
        }
      }
      bool bad;
      bad = separator_core<NX, NU, STRICT, KEEP, 16>(lane, ab, xc, sout, Lrow,
                                                     pv[(s - wgbase) >> 1].ab[(s - wgbase) & 1], WP,
                                                     (!KEEP && (recout & 2)) ? Fblk(F, d, b, l, s + 1) : nullptr);
      if (bad && lane == 0) flag_failure(info, d, b);
      if (KEEP && lane < NX) store_row<NX>(Fblk(F, d, b, l, s + 1) + gi * NX, Lrow);
      if constexpr (!STRICT) {
        if (recout & 1) {  // record f_a | f_bb | z_sep of this separator (layout of separator_wave)
          double* myrec = rec + ((size_t)b * N + s) * (2 * NX * NX + NX);
          const int grp = lane / NX;
          if (grp < 2) {
            if ((grp == 0 ? a : bb) >= 0) {
              double row[NX];
#pragma unroll
              for (int c = 0; c < NX; ++c) row[c] = sout.X[gi * LD + grp * NX + c];
              store_row<NX>(myrec + grp * NX * NX + gi * NX, row);
            }
          } else if (grp == 2) {
            myrec[2 * NX * NX + gi] = sout.X[gi * LD + 2 * NX];
          }
        }
      }
    }
    SEG(23);
    __syncthreads();
    SEG(24);

    // Schur update of my two knots, then rotate the column roles
    const double* fa = sout.X;           // f_a(k, c)  = X[k * LD + c]
    const double* fb = sout.X + NX;      // f_bb(k, c) = X[k * LD + NX + c]
    const double* zsp = sout.X + 2 * NX; // z_sep(k)   = X[k * LD + 2 NX]
    if (KEEP && has_knot && active) store_row<NX>(Fblk(F, d, b, l, i) + r * NX, E);
    {
      const bool leftchild = (base & T) == 0;
      row_update_live<NX, LD, STRICT>(E, C, zz, left ? fa : fb, left ? fb : fa, zsp, left ? a >= 0 : bb >= 0,
                                      left ? bb >= 0 : a >= 0, active, !active && i == s + 1, r,
                                      left == leftchild);
    }
    SEG(25);
    __syncthreads();  // xs / pv are reused by the next level
    SEG(26);
  }

  // ---- hand-off: column JB, the live outer column at level JB, the rhs block
  bool handoff = has_knot;
  if constexpr (!STRICT && !KEEP) { if (lean) handoff = has_knot && (i == wgbase || i == wgbase + NK - 1); }
  if (handoff) {
    const int l = JB;
    const int base = (i >> (l + 1)) << (l + 1), s = base + (1 << l) - 1;
    int a, bb;
    outer_columns(base, l, N, a, bb);
    const bool calc_lambda = (i == 0) || (i & ((1 << l) - 1)) != 0;
    // lambda rows of column JB that are still structural zeros / hold a Cholesky factor written
    // by a later separator are not touched (same rule as the level kernels)
    const bool real_row = !lam || calc_lambda || ((i & ((1 << l) - 1)) == 0 && false);
    if (real_row || lam) {
      // E: every row is meaningful here except lambda rows not yet eliminated (zeros) -- writing
      // the zeros is harmless and keeps re-solves free of stale data
      store_row<NX>(Fblk(F, d, b, l, i) + r * NX, E);
    }
    const int cc = (i <= s) ? a : bb;
    if (cc >= 0) store_row<NX>(Fblk(F, d, b, cc, i) + r * NX, C);
    z[((size_t)b * N + i) * ROWS + r] = zz;
  }
#ifdef NDLQR_SEGTIME
  __builtin_amdgcn_s_waitcnt(0);
#endif
  SEG(27);
}

}  // namespace ndlqr
