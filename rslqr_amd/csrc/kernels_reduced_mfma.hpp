// kernels_reduced_mfma.hpp -- the separator-only ("reduced") schedule for blocks that fill 16x16
// matrix-core tiles (nstates a multiple of 16, nstates + ninputs of 4: BASELINE.json config 5's
// (64,16)). Fast mode without KEEP. One kernel, launched once per tree level; no factor array, no
// knot states, no leaf pass, no Schur pass.
//
// DESIGN.md section 3.1 (same algebra as kernels_bottom_reduced.hpp, which serves 6 <= n <= 15):
// eliminating the states and inputs of every knot (ndlqr_SolveLeaf, src/nested_dissection.c:10-105)
// leaves a block-tridiagonal system in the multipliers,
//     leafS_s = [A_s | B_s] diag(1/Q_s, 1/R_s) [A_s | B_s]' + Q_{s+1}^-1
//     r_a = -A_s Q_s^-1  (coupling to s-1)      r_bb = -Q_{s+1}^-1 A_{s+1}'  (coupling to s+1)
//     leafb_s = [A_s | B_s] z(s).xu - z(s+1).lambda - z(s+1).x
// and the nested-dissection levels (ndlqr_FactorInnerProduct nested_dissection.c:114-134, the Cholesky
// of src/solve.c:87-98, ndlqr_SolveCholeskyFactor :136-152, ndlqr_UpdateShurFactor :154-171) are block
// cyclic reduction on it. Separator s of level l with the subtree [base, base + 2^(l+1)), neighbours
// A = base - 1 and B = base + 2^(l+1) - 1 (both of a higher level):
//     S-bar = leafS - DL - DR      R = [r_a | r_bb | b~],  r_a = -CA, r_bb = -CB (level 0: from the data),
//                                  b~ = leafb - gL - gR
//     X = S-bar^-1 R = [f_a | f_bb | z_sep]                  -> record of s (back-substitution)
//     DR[A] += r_a' f_a    gR[A] += r_a' z_sep    DL[B] += r_bb' f_bb    gL[B] += r_bb' z_sep
//     left child of B:  CA[B] = f_bb' r_a          right child of A:  CB[A] = r_a' f_bb
// Every slot block has ONE writer per launch (a separator has one left and one right neighbour, a
// neighbour one adjacent subtree per level and side), so the pushes are plain read-modify-writes;
// the level-0 launch reaches DL, DR, gL, gR of every separator of a higher level and stores instead
// of adding: nothing has to be zeroed between solves.
//
// Slots: separators of level >= 1 are the odd ones; slot of s at index s >> 1,
//     DL | DR | CA | CB (n x n, row-major) | gL | gR (n)  =  4 n^2 + 2 n doubles.
#pragma once
#include "kernels_mfma.hpp"

namespace ndlqr {

__device__ __forceinline__ double* reduced_slot(double* red, const Dims& d, const int b, const int s) {
  return red + ((size_t)b * (d.N >> 1) + (s >> 1)) * (4 * (size_t)d.n * d.n + 2 * d.n);
}

// pitch of the staged [A_s | B_s] rows in LDS: = 4 (mod 8) doubles, so that the sixteen rows a
// matrix-core operand fetch touches (lane li: row, lk: four consecutive doubles) fall into disjoint banks
__host__ __device__ inline int reduced_stage_pitch(const int w) { return (w % 8 == 4) ? w : w + 4; }

// NB = n / 16, NTHR threads (a multiple of 64), CT column tiles of the panel resident at a time.
//   grid (N >> (l+1), batch), block NTHR;  host (plan_reduced_generic): NTHR / 64 >= NB,
//   dynamic LDS = n (n + 1) + n (16 min(CT, 2 NB + 1) + 1) + 17 n + 2 (n + m) + 2 n doubles; the staged
//   [A_s | B_s] (n rows of reduced_stage_pitch(w)) and later r_a | r_bb (2 n (n + 1)) lie over the first arrays.
// LEVEL0: the launch of tree level 0 (couplings from the problem data, pushes are stores).
//
// Written for memory-level parallelism: every global operand is requested as early as its address is
// known and consumed as late as possible (the slot blocks DL, DR, gL, gR and the first panel chunk at
// kernel entry, the next panel chunk under the products of the current one), loads are unconditional
// on clamped indices and LDS stores likewise (a store under a lane predicate makes the compiler sink its
// load behind the predicate, and the loads then complete one after the other), and the matrix-core
// loops fetch the operand fragments of several k-steps before the first product.
template <int NB, int NTHR, int CT, bool LEVEL0>
__global__ __launch_bounds__(NTHR, 4) void separator_reduced_mfma(Dims d, int l, const double* __restrict__ AB,
                                                                  const double* __restrict__ QR,
                                                                  const double* __restrict__ rhs, double* red,
                                                                  double* __restrict__ rec, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int n = 16 * NB, nn = n * n, ns = n + 1, tiles = NB, ctl = 2 * NB + 1;  // column tiles: f_a, f_bb, [z_sep | padding]
  constexpr int ctc = ctl < CT ? ctl : CT, xs = 16 * ctc + 1, NW = NTHR / 64;
  constexpr int MAXS = (NB * NB + NW - 1) / NW;        // S-bar tiles per wavefront
  constexpr int PT = (16 * n + NTHR - 1) / NTHR;       // panel elements per thread and column tile
  constexpr int PR = n + 1;                            // pitch of r_a, r_bb in the push phase
  static_assert(NW >= NB && NB * ctc <= kSepMaxPanelTiles * NW && n / 4 <= 16, "work distribution of the shared phases");
  const int w = d.w, N = d.N, rows = d.rows;
  const int b = blockIdx.y;
  const int T = 2 << l, base = blockIdx.x * T, s = base + (1 << l) - 1;
  const bool hasA = base > 0, hasB = base + T < N, leftchild = (base & T) == 0, first = s == 0;
  double* S = sm;
  double* X = S + n * ns;
  double* Wd = X + n * xs;   // NB blocks of 16 x 17: inverses of the diagonal blocks of L
  double* dq = Wd + n * 17;  // 1 / [Q_s | R_s]  (state entries of knot 0: zero -- its state is fixed)
  double* zc = dq + w;       // rhs(s).xu scaled likewise (state entries of knot 0: -x0)
  double* q1 = zc + w;       // 1 / Q_{s+1}
  double* bz = q1 + n;       // b~
  double* stage = sm;        // [A_s | B_s], pitch P, over S and X until S-bar is formed
  const int P = reduced_stage_pitch(w);
  const int tid = threadIdx.x;
  // (the wavefront index as a scalar: everything that depends on it branches uniformly)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const SepGeom geo = {n, ns, xs, tiles, lane, wave, NW, li, lk};
  SEG_INIT();

  const double* ab = AB + ((size_t)b * N + s) * n * w;   // [A_s | B_s]
  const double* ab1 = ab + (size_t)n * w;                 // [A_{s+1} | B_{s+1}]  (s + 1 <= N - 1)
  const double* qr = QR + ((size_t)b * N + s) * w;
  const double* r0 = rhs + ((size_t)b * N + s) * rows;
  const double* myslot = reduced_slot(red, d, b, LEVEL0 ? 1 : s);  // (not read at level 0)
  double* slotA = reduced_slot(red, d, b, hasA ? base - 1 : 1);
  double* slotB = reduced_slot(red, d, b, hasB ? base + T - 1 : 1);
  double* myrec = rec + ((size_t)b * N + s) * (2 * (size_t)nn + n);
  const int ksteps = w / 4;

  // ---- panel elements: thread -> (row i, column cl) of a column tile. Level 0 takes r_bb from the rows of
  //      A_{s+1} (sixteen consecutive row indices = one 128-byte line of a row of A_{s+1} per sixteen lanes).
  //      fetch = the raw global operand (uniform branches only), write = what enters the panel.
  auto panel_index = [&](const int gt, const int e, int& i, int& cl) {
    if (LEVEL0 && gt >= tiles && gt < 2 * tiles) { i = ((e >> 8) << 4) | (e & 15); cl = (e >> 4) & 15; }
    else { i = e >> 4; cl = e & 15; }
  };
  auto panel_fetch = [&](const int t0, double (&pf)[ctc * PT]) {
#pragma unroll
    for (int t = 0; t < ctc; ++t) {
      const int gt = t0 + t;
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        const int e = tid + u * NTHR, ec = e < 16 * n ? e : 16 * n - 1;
        int i, cl;
        panel_index(gt, ec, i, cl);
        double v = 0.0;
        if (gt < tiles) {
          if (hasA) v = LEVEL0 ? ab[(size_t)i * w + 16 * gt + cl] : myslot[2 * nn + i * n + 16 * gt + cl];
        } else if (gt < 2 * tiles) {
          const int c = 16 * (gt - tiles) + cl;
          if (hasB) v = LEVEL0 ? ab1[(size_t)c * w + i] : myslot[3 * nn + i * n + c];
        }
        pf[t * PT + u] = v;
      }
    }
  };
  auto panel_write = [&](const int t0, const double (&pf)[ctc * PT]) {
#pragma unroll
    for (int t = 0; t < ctc; ++t) {
      const int gt = t0 + t;
      if (gt >= ctl) continue;
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        const int e = tid + u * NTHR, ec = e < 16 * n ? e : 16 * n - 1;  // (surplus threads rewrite the last element)
        int i, cl;
        panel_index(gt, ec, i, cl);
        const double raw = pf[t * PT + u];
        double v;
        if (gt < tiles) v = LEVEL0 ? -raw * dq[16 * gt + cl] : -raw;
        else if (gt < 2 * tiles) v = LEVEL0 ? -raw * q1[i] : -raw;
        else v = cl == 0 ? bz[i] : 0.0;
        X[i * xs + 16 * t + cl] = v;
      }
    }
  };

  // ---- requests at kernel entry: [A_s | B_s] and the weights / rhs of knots s, s + 1
  const int words = n * w / 2;               // 16-byte words of [A_s | B_s]
  const float inv_w = 1.0f / (float)w;
  const int kc = tid < w ? tid : w - 1, ic = tid < n ? tid : n - 1;
  const double qv = qr[kc], q1v = qr[w + ic], rxu = r0[n + kc], rl0 = r0[ic], za = r0[rows + ic], zb = r0[rows + n + ic];
  // ---- stage [A_s | B_s] (rows of pitch P; n = 64, w = 80: one round) and the diagonal weights / rhs of knots s, s + 1
  constexpr int SG = 5;
  for (int e0 = 0; e0 < words; e0 += SG * NTHR) {
    double2 st[SG];
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const int e = e0 + tid + u * NTHR;
      st[u] = reinterpret_cast<const double2*>(ab)[e < words ? e : words - 1];
    }
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const int e = e0 + tid + u * NTHR, ec = e < words ? e : words - 1;
      const int row = (int)(((float)(2 * ec) + 0.5f) * inv_w), col = 2 * ec - row * w;
      *reinterpret_cast<double2*>(stage + row * P + col) = st[u];
    }
  }
  {
    const bool fx = first && kc < n;
    const double inv = 1.0 / qv;
    dq[kc] = fx ? 0.0 : inv;
    zc[kc] = fx ? -rl0 : rxu * inv;
    q1[ic] = 1.0 / q1v;
  }
  // requests of the later phases (the staging registers are free again): DL + DR of this wavefront's S-bar
  // tiles and gL + gR (consumed behind the S-bar products), the first panel chunk
  double dlr[MAXS][8];
#pragma unroll
  for (int idx = 0; idx < MAXS; ++idx) {
    const int item = wave + idx * NW, itc = item < NB * NB ? item : NB * NB - 1, rt = itc / NB, ct = itc % NB;
    const double* p = myslot + (size_t)(16 * rt + lk) * n + 16 * ct + li;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      dlr[idx][gg] = LEVEL0 ? 0.0 : p[4 * gg * n];
      dlr[idx][4 + gg] = LEVEL0 ? 0.0 : p[nn + 4 * gg * n];
    }
  }
  double gl = 0.0, gr = 0.0;
  if (!LEVEL0) { gl = myslot[4 * nn + ic]; gr = myslot[4 * nn + n + ic]; }
  double pf[ctc * PT];
  panel_fetch(0, pf);

  __syncthreads();
  SEG(50);

  // ---- leafS = [A_s | B_s] diag(dq) [A_s | B_s]' (- DL - DR + Q_{s+1}^-1 at the write-out) as matrix-core tiles, held in
  //      the accumulators until every wavefront has finished reading the staged block (S-bar goes over it)
  mfma_acc_t sacc[MAXS];
  {
    const double* arow[MAXS];
    const double* brow[MAXS];
#pragma unroll
    for (int idx = 0; idx < MAXS; ++idx) {
      const int item = wave + idx * NW, itc = item < NB * NB ? item : NB * NB - 1, rt = itc / NB, ct = itc % NB;
      arow[idx] = stage + (16 * rt + li) * P + lk;
      brow[idx] = stage + (16 * ct + li) * P + lk;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) sacc[idx][gg] = 0.0;
    }
    constexpr int CH = 4;
    for (int q0 = 0; q0 < ksteps; q0 += CH) {
      double af[MAXS][CH], bf[MAXS][CH], dv[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = q0 + c < ksteps ? q0 + c : ksteps - 1;
        dv[c] = q0 + c < ksteps ? dq[4 * q + lk] : 0.0;  // surplus steps of the last round multiply by zero
#pragma unroll
        for (int idx = 0; idx < MAXS; ++idx) { af[idx][c] = arow[idx][4 * q]; bf[idx][c] = brow[idx][4 * q]; }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int idx = 0; idx < MAXS; ++idx)
          sacc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[idx][c], bf[idx][c] * dv[c], sacc[idx], 0, 0, 0);
    }
  }
  // b~ = [A_s | B_s] zc - z(s+1).lambda - z(s+1).x / Q_{s+1} - gL - gR   (threads < n; the others repeat row n - 1)
  double bt;
  {
    const double* arow = stage + ic * P;
    double acc = -fma(zb, q1[ic], za);
    for (int k0 = 0; k0 < w; k0 += 8) {
      double av[8], zv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u < w ? k0 + u : w - 1;
        av[u] = arow[k];
        zv[u] = k0 + u < w ? zc[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fma(av[u], zv[u], acc);
    }
    bt = acc;
  }
  __syncthreads();
  SEG(51);
  bz[ic] = bt - (gl + gr);
  if (ctl > ctc) panel_write(0, pf);  // the staged block is dead: the first chunk (no b~ column in it) goes into the panel
#pragma unroll
  for (int idx = 0; idx < MAXS; ++idx) {
    const int item = wave + idx * NW;
    if (item < NB * NB) {
      const int rt = item / NB, ct = item % NB;
      double* dst = S + (16 * rt + lk) * ns + 16 * ct + li;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const int i = 16 * rt + lk + 4 * gg, j = 16 * ct + li;
        dst[4 * gg * ns] = sacc[idx][gg] + (i == j ? q1[i] : 0.0) - (dlr[idx][gg] + dlr[idx][4 + gg]);
      }
    }
  }
  __syncthreads();
  SEG(52);

  sep_cholesky(geo, S, Wd, info, d, b);
  SEG(53);
  sep_invert(geo, S, Wd);
  SEG(54);

  // ---- the panel, CT column tiles at a time: build, X = W' (W R), record
  for (int t0 = 0; t0 < ctl; t0 += ctc) {
    const int tc = ctl - t0 < ctc ? ctl - t0 : ctc;
    if (t0 > 0 || ctl <= ctc) panel_write(t0, pf);  // (the first chunk of a chunked panel is in place already)
    // the next chunk: in flight under the products of this one (not earlier: registers that wait for a load
    // across the Cholesky get spilled, and the spill waits for the load)
    if (t0 + ctc < ctl) panel_fetch(t0 + ctc, pf);
    if (t0 > 0 || ctl <= ctc) __syncthreads();
    SEG(55);
    sep_panel_solve(geo, S, Wd, X, tc);
    SEG(56);
    // record f_a | f_bb | z_sep (what the back-substitution reads -- and the push phase below): the chunk's
    // elements dealt to the threads, LDS reads first, then the stores
    {
      constexpr int RE = (16 * ctc * n + NTHR - 1) / NTHR;  // element e: column e & 15 of tile (e >> 4) / n, row (e >> 4) % n
      double v[RE];
#pragma unroll
      for (int u = 0; u < RE; ++u) {
        const int e = tid + u * NTHR, ec = e < 16 * tc * n ? e : 16 * tc * n - 1;
        const int t = (ec >> 4) / n, i = (ec >> 4) % n;
        v[u] = X[i * xs + 16 * t + (ec & 15)];
      }
#pragma unroll
      for (int u = 0; u < RE; ++u) {
        const int e = tid + u * NTHR;
        const int t = (e >> 4) / n, i = (e >> 4) % n, gt = t0 + t, cl = e & 15;  // (t is uniform over a wavefront: 64 | 16 n)
        if (e < 16 * tc * n) {
          if (gt < tiles) { if (hasA) myrec[i * n + 16 * gt + cl] = v[u]; }
          else if (gt < 2 * tiles) { if (hasB) myrec[nn + i * n + 16 * (gt - tiles) + cl] = v[u]; }
          else if (cl == 0) myrec[2 * nn + i] = v[u];
        }
      }
    }
    __syncthreads();  // the next chunk overwrites the panel
    SEG(57);
  }

  // ---- pushes. S-bar / W, the panel chunk and the diagonal-block inverses are dead: r_a and r_bb (n x n each,
  //      pitch n + 1) take their place in LDS and serve as matrix-core operands; the solved blocks come back
  //      from the record this workgroup has just written (L2), all k-steps of a column tile per request round.
  //      unit (X column tile, coupling block) -> NB output tiles:
  //        [0, NB)        f_a tile,  r_a:   DR[A] += r_a' f_a
  //        [NB, 2 NB)     f_bb tile, r_bb:  DL[B] += r_bb' f_bb
  //        [2 NB, 3 NB)   f_bb tile, r_a:   CA[B] = f_bb' r_a (left child)  or  CB[A] = r_a' f_bb
  //        3 NB           z_sep, r_a:       gR[A] += r_a' z_sep
  //        3 NB + 1       z_sep, r_bb:      gL[B] += r_bb' z_sep
  double* Ra = sm;
  double* Rb = sm + n * PR;
  {
    constexpr int RS = (nn + NTHR - 1) / NTHR;
    double ta[RS], tb[RS];
#pragma unroll
    for (int u = 0; u < RS; ++u) {
      const int e = tid + u * NTHR, ec = e < nn ? e : nn - 1, hi = ec / n, lo = ec % n;
      ta[u] = 0.0; tb[u] = 0.0;
      if (hasA) ta[u] = LEVEL0 ? ab[(size_t)hi * w + lo] : myslot[2 * nn + ec];
      // level 0: r_bb(k, j) = -A_{s+1}(j, k) / Q_{s+1}(k): walk the rows of A_{s+1} (k fastest)
      if (hasB) tb[u] = LEVEL0 ? ab1[(size_t)hi * w + lo] : myslot[3 * nn + ec];
    }
#pragma unroll
    for (int u = 0; u < RS; ++u) {
      const int e = tid + u * NTHR, ec = e < nn ? e : nn - 1, hi = ec / n, lo = ec % n;
      Ra[hi * PR + lo] = LEVEL0 ? -ta[u] * dq[lo] : -ta[u];
      if (LEVEL0) Rb[lo * PR + hi] = -tb[u] * q1[lo];
      else Rb[hi * PR + lo] = -tb[u];
    }
  }
  __threadfence_block();
  __syncthreads();
  SEG(58);
  for (int unit = wave; unit < 3 * NB + 2; unit += NW) {
    const int grp = unit < 3 * NB ? unit / NB : 3 + (unit - 3 * NB);  // 0 DR, 1 DL, 2 coupling, 3 gR, 4 gL
    const int ct = unit < 3 * NB ? unit % NB : 0;
    const bool use_ra = grp == 0 || grp == 2 || grp == 3;
    if (use_ra && !hasA) continue;
    if ((grp == 1 || grp == 2 || grp == 4) && !hasB) continue;
    const bool vec = grp >= 3;
    const bool xa = grp == 2 && leftchild;  // the solved tile is the A operand (CA = f_bb' r_a)
    const double* Rl = (use_ra ? Ra : Rb) + lk * PR + li;
    // destination: tile (orow, ocol) of an n x n block (row pitch n), or sixteen entries of a vector
    double* dstb = grp == 0 ? slotA + nn : grp == 1 ? slotB : grp == 2 ? (leftchild ? slotB + 2 * nn : slotA + 3 * nn)
                 : grp == 3 ? slotA + 4 * nn + n : slotB + 4 * nn;
    const bool accumulate = !LEVEL0 && grp != 2;
    const int pstep = vec ? 4 : 4 * n;
    // operand fragments of the solved column tile (all n / 4 k-steps) and the tiles to be updated: one request round
    double xf[n / 4];
    mfma_acc_t acc[NB];
    {
      const double* xsrc = vec ? myrec + 2 * nn + lk : myrec + (grp == 0 ? 0 : nn) + (size_t)lk * n + 16 * ct + li;
#pragma unroll
      for (int q = 0; q < n / 4; ++q) xf[q] = vec ? xsrc[4 * q] : xsrc[(size_t)4 * q * n];
#pragma unroll
      for (int rt = 0; rt < NB; ++rt) {
        const int orow = xa ? ct : rt, ocol = xa ? rt : ct;
        const double* p = vec ? dstb + 16 * rt + lk : dstb + (size_t)(16 * orow + lk) * n + 16 * ocol + li;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) acc[rt][gg] = accumulate ? p[gg * pstep] : 0.0;
      }
      if (vec) {
#pragma unroll
        for (int q = 0; q < n / 4; ++q) xf[q] = li == 0 ? xf[q] : 0.0;
      }
    }
#pragma unroll
    for (int rt = 0; rt < NB; ++rt) {
      const double* rl = Rl + 16 * rt;
      double rv[n / 4];
#pragma unroll
      for (int q = 0; q < n / 4; ++q) rv[q] = rl[4 * q * PR];
      if (xa) {
#pragma unroll
        for (int q = 0; q < n / 4; ++q) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[q], rv[q], acc[rt], 0, 0, 0);
      } else {
#pragma unroll
        for (int q = 0; q < n / 4; ++q) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[q], xf[q], acc[rt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < NB; ++rt) {
      const int orow = xa ? ct : rt, ocol = xa ? rt : ct;
      double* p = vec ? dstb + 16 * rt + lk : dstb + (size_t)(16 * orow + lk) * n + 16 * ocol + li;
      if (!vec || li == 0) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) p[gg * pstep] = acc[rt][gg];
      }
    }
  }
  SEG(59);
}

}  // namespace ndlqr
