// kernels_reduced_mfma.hpp -- the separator-only ("reduced") schedule for every block size up to 64 states
// that has no size-specialised instance, on 16x16 matrix-core tiles (BASELINE.json config 5's (64,16) fills
// them; other sizes are zero-padded in LDS). Fast mode without KEEP_FACT. One kernel, launched once per tree
// level; no factor array, no knot states, no leaf pass, no Schur pass. With NDLQR_FLAG_KEEP_RECORDS the sweep
// also keeps the Cholesky factor of every separator, and rhs_reduced_generic (below) re-solves for new right-hand sides.
//
// DESIGN.md section 3.5a (same algebra as kernels_bottom_reduced.hpp, which serves 6 <= n <= 15):
// eliminating the states and inputs of every knot (ndlqr_SolveLeaf, src/nested_dissection.c:10-105)
// leaves a block-tridiagonal system in the multipliers,
//     leafS_s = [A_s | B_s] diag(1/Q_s, 1/R_s) [A_s | B_s]' + Q_{s+1}^-1
//     r_a = -A_s Q_s^-1  (coupling to s-1)      r_bb = -Q_{s+1}^-1 A_{s+1}'  (coupling to s+1)
//     leafb_s = [A_s | B_s] z(s).xu - z(s+1).lambda - z(s+1).x
// and the nested-dissection levels (ndlqr_FactorInnerProduct nested_dissection.c:114-134, the Cholesky
// of src/solve.c:87-98, ndlqr_SolveCholeskyFactor :136-152, ndlqr_UpdateShurFactor :154-171) are block
// cyclic reduction on it. Separator s of level l with the subtree [base, base + 2^(l+1)), neighbours
// A = base - 1 and B = base + 2^(l+1) - 1 (both of a higher level):
//     S-bar = leafS - DL - DR = L L'   R = [r_a | r_bb | b~],  r_a = -CA, r_bb = -CB (level 0: from the data),
//                                      b~ = leafb - gL - gR
//     Y = L^-1 R = [Y_a | Y_b | y~]    (X = S-bar^-1 R = [f_a | f_bb | z_sep] of the reference is never formed)
//     DR[A] += Y_a'Y_a    gR[A] += Y_a'y~    DL[B] += Y_b'Y_b    gL[B] += Y_b'y~      (= r_a' f_a, r_a' z_sep, ..)
//     left child of B:  CA[B] = Y_b'Y_a          right child of A:  CB[A] = Y_a'Y_b   (= f_bb' r_a, r_a' f_bb)
//     record of s: L (the inverses of its diagonal 16x16 blocks in their place), packed lower triangle, and y~; the
//     back-substitution solves  y_s = L^-T (y~ - L^-1 (r_a y_A + r_bb y_B))  with the couplings from where this
//     kernel read them (slot / problem data)
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
__host__ __device__ constexpr int reduced_stage_pitch(const int w) { return (w % 8 == 4) ? w : w + 4; }
// pitch of S-bar / L (n = 16 NB): the same rule -- its blocks are fetched as A operands (lane li: row) by every product
// of the Cholesky and of the panel substitution; n + 1 would put four lanes on a bank
__host__ __device__ constexpr int reduced_s_pitch(const int n) { return reduced_stage_pitch(n); }
// pitch of the solved panel halves Y in the push phase, = 16 (mod 32) doubles: they are written from accumulator tiles
// and fetched as the transposed operand (lane lk: row, li: column), and lk * pitch + li then covers every bank twice
__host__ __device__ constexpr int reduced_y_pitch(const int n) { return n % 32 == 16 ? n : n + 16; }

// NB = ceil(n / 16), NTHR threads (a multiple of 64, at least 64 NB and max(16 NB, n + m rounded up to 4)).
//   grid (N >> (l+1), batch), block NTHR;
//   dynamic LDS = reduced_lds_doubles(padded n, padded w) doubles: the weights / rhs arrays, then a region that
//   first holds the staged [A_s | B_s] (n rows of reduced_stage_pitch(w)), then S-bar / L (n rows of reduced_s_pitch(n)),
//   then the solved panel halves Y_a, Y_b (n rows of reduced_y_pitch(n)) for the pushes.
// LEVEL0: the launch of tree level 0 (couplings from the problem data, pushes are stores). PAD: see below.
// NTHR = 64 NB ("two rounds"): every wavefront solves column tile c of r_a AND of r_bb; the pushes run in two rounds on
// ONE panel array in LDS (Y_a, then Y_b, both from the registers of phase B) -- 45 KB at (64,16): THREE workgroups
// share a CU. NTHR = 128 NB (inputs too wide for the lanes of 64 NB threads): Y_b in its own array.
//
// Phases (workgroup barriers only between them):
//   A  stage [A_s | B_s]; leafS tiles on the matrix cores; S-bar = leafS - DL - DR + Q_{s+1}^-1 into LDS; blocked
//      Cholesky with look-ahead (reduced_cholesky below); beside it y~ = L^-1 b~ (vector ALU) and the record stores
//   B  the 2 NB column tiles of [r_a | r_bb] are dealt to the wavefronts. A wavefront takes the operand fragments of
//      its tile straight from global memory and solves Y = L^-1 R by block forward substitution in the accumulators
//      -- an accumulator tile is at once the B operand of the next product, so nothing goes through LDS and nobody
//      waits for anybody -- and KEEPS Y in registers
//   C  Y_a into LDS over the dead S-bar / L, from the registers (Y_b: its own array, or the second round)
//   D  pushes: the wavefront that holds column tile c of Y_a forms the tiles of DR[A](:, c) = Y_a' Y_a(:, c) on and
//      below the diagonal (DR, DL and S-bar are symmetric: their upper tiles are neither computed, stored nor
//      read anywhere) and the coupling tiles (c, j <= c) as Y_a' Y_b; the one that holds tile c of Y_b forms the lower
//      tiles of DL[B](:, c) and the coupling tiles (i < c, c): NB + 1 and NB tile products, one operand from registers
// Written for memory-level parallelism: loads are unconditional on clamped indices and requested as early as
// their address is known, LDS stores likewise (a store under a lane predicate makes the compiler sink its load
// behind the predicate, and the loads then complete one after the other).

// two_round: one wavefront per tile column (NTHR = 64 NB): r_a and r_bb take turns in ONE array over the dead
// S-bar / W (no separate r_bb array)
__host__ __device__ inline int reduced_lds_doubles(const int n, const int w, const bool two_round) {
  // S-bar / L, then Y of r_a (two rounds: and of r_bb) over it; otherwise Y of r_bb behind the two
  int later = n * reduced_s_pitch(n) > n * reduced_y_pitch(n) ? n * reduced_s_pitch(n) : n * reduced_y_pitch(n);
  if (!two_round) later += n * reduced_y_pitch(n);
  int big = n * reduced_stage_pitch(w);  // staged [A_s | B_s]: dead before any of them
  if (later > big) big = later;
  return w + (w > n ? w : n) + 2 * n + big;  // dq (w), zc (later scratch of the substitutions: max(w, n)), q1, b~ (n each)
}

// lower-triangle tile t -> its block row (block column: t - row (row + 1) / 2)
__device__ __forceinline__ int tri_row(const int t) {
  int r = 0;
  while ((r + 1) * (r + 2) / 2 <= t) ++r;
  return r;
}

// ------------------------------------------------------------------------------------- block substitutions (vector ALU)
// One block step of L v = t (forward) or L'y = v (backward) for ONE vector, by ONE wavefront, with what the blocked
// Cholesky leaves behind: Lo(i, k) = L(i, k) for (i, k) in different 16-blocks (i > k), Di(i, k) = entry (i, k) of the
// inverse of the diagonal block that holds both (i >= k). Lane = (row r of the block, quarter `seg` of the summation
// index); `vec` (LDS, n entries) holds the right-hand side and receives the solution block by block; `tmp`: 16 doubles
// of LDS. n need not be a multiple of 16 (the last block is then short). The wavefront's own LDS accesses complete in
// order; the fences only keep the compiler from moving them across.
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double quad_sum(double v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}
template <class LoAt, class DiAt>
__device__ __forceinline__ void tri_forward_block(const int ib, const int n, double* vec, double* tmp, const int lane,
                                                  LoAt Lo, DiAt Di) {
  const int r = lane >> 2, seg = lane & 3, i = 16 * ib + r, ic = i < n ? i : n - 1;
  double acc = 0.0;
  for (int k0 = 0; k0 < 16 * ib; k0 += 16) {
    double lv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { lv[u] = Lo(ic, k0 + seg + 4 * u); vv[u] = vec[k0 + seg + 4 * u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fma(lv[u], vv[u], acc);
  }
  const double t = vec[ic] - quad_sum(acc);
  if (seg == 0) tmp[r] = t;
  wave_lds_order();
  double a2 = 0.0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = seg + 4 * u, cc = c <= r ? c : r;
    const int col = 16 * ib + cc < ic ? 16 * ib + cc : ic;
    a2 = fma(c <= r ? Di(ic, col) : 0.0, tmp[cc], a2);
  }
  a2 = quad_sum(a2);
  if (seg == 0 && i < n) vec[i] = a2;
  wave_lds_order();
}
template <class LoAt, class DiAt>
__device__ __forceinline__ void tri_backward_block(const int ib, const int n, double* vec, double* tmp, const int lane,
                                                   LoAt Lo, DiAt Di) {
  const int r = lane >> 2, seg = lane & 3, i = 16 * ib + r, ic = i < n ? i : n - 1;
  double acc = 0.0;
  for (int k0 = 16 * (ib + 1); k0 < n; k0 += 16) {
    double lv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + seg + 4 * u, kc = k < n ? k : n - 1;
      lv[u] = Lo(kc, ic);
      vv[u] = k < n ? vec[kc] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fma(lv[u], vv[u], acc);
  }
  const double t = vec[ic] - quad_sum(acc);
  if (seg == 0) tmp[r] = t;
  wave_lds_order();
  double a2 = 0.0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = seg + 4 * u;
    const bool ok = c >= r && 16 * ib + c < n;
    const int cc = ok ? c : r, row = ok ? 16 * ib + c : ic;
    a2 = fma(ok ? Di(row, ic) : 0.0, tmp[cc], a2);
  }
  a2 = quad_sum(a2);
  if (seg == 0 && i < n) vec[i] = a2;
  wave_lds_order();
}

// ------------------------------------------------------------------------------------- blocked Cholesky, in place
// Lower Cholesky of one 16 x 16 diagonal block by ONE wavefront (rb_chol_inv, kernels_dpp.hpp) that leaves the INVERSE
// of the block's factor in its place (row pitch ns; zeros above the diagonal): nothing after this step wants the
// diagonal block of L itself -- the panel below it, the substitutions and the records all work with the inverse.
// Every lane stores (lanes >= 16, replicas, into the column behind the block -- the tile above the diagonal, never
// read, or the pad column): a store under a lane predicate makes the compiler sink the recurrence behind it.
__device__ __forceinline__ bool chol16_inverse_in_place(double* Sblk, const int ns, const int lane) {
  const int r = lane & 15;
  double acc[16], w[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = Sblk[r * ns + c];
  const bool bad = rb_chol_inv<16>(r, acc, w);
  const int wc = lane < 16 ? lane : 16;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) Sblk[rr * ns + wc] = w[rr];
  return bad;
}

// Blocked Cholesky of S (n = 16 NB, lower block triangle, row pitch ns) by NW wavefronts, with look-ahead: per block
// column two workgroup barriers -- behind the panel L21 = A21 D^-T and behind the trailing update -- and the first
// wavefront takes the NEXT diagonal tile through its update and straight on into its factorisation while the others
// finish the trailing update. On return S holds L below the diagonal blocks and the inverses of the diagonal blocks
// of L in their place. side(jb, wave): called by every wavefront but the first (by the only one when NW = 1) once block
// row jb is final (its L blocks and the inverse of its diagonal block), beside the factorisation of the next diagonal
// block: work that would otherwise wait at the barrier.
template <int NB, int NW, class Side>
__device__ __forceinline__ void reduced_cholesky(double* S, const int ns, const int lane, const int wave, int* __restrict__ info,
                                                 const Dims& d, const int b, Side side) {
  const int li = lane & 15, lk = lane >> 4;
  auto diag = [&](const int jb) {
    const bool bad = chol16_inverse_in_place(S + 16 * jb * ns + 16 * jb, ns, lane);
    if (bad && lane == 0) flag_failure(info, d, b);
  };
  // tile (it, ct) -= L(it, jb) L(ct, jb)'
  auto trailing_tile = [&](const int jb, const int it, const int ct) {
    double* Ct = S + (16 * it + lk) * ns + 16 * ct + li;
    double a[4], bl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = -S[(16 * it + li) * ns + 16 * jb + 4 * q + lk];
      bl[q] = S[(16 * ct + li) * ns + 16 * jb + 4 * q + lk];
    }
    mfma_acc_t acc = {Ct[0], Ct[4 * ns], Ct[8 * ns], Ct[12 * ns]};
    acc = mfma4(a, bl, acc);
    Ct[0] = acc[0]; Ct[4 * ns] = acc[1]; Ct[8 * ns] = acc[2]; Ct[12 * ns] = acc[3];
  };
  if (wave == 0) diag(0);
  __syncthreads();
#pragma unroll
  for (int jb = 0; jb < NB; ++jb) {
    const int rem = NB - 1 - jb;
    if (rem == 0) {
      if (NW == 1 || wave != 0) side(jb, wave);
      break;
    }
    for (int it = jb + 1 + wave; it < NB; it += NW) {  // L21 = A21 D^-T (D^-1: the inverse in the diagonal block)
      double a[4], bw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[q] = S[(16 * it + li) * ns + 16 * jb + 4 * q + lk];
        bw[q] = S[(16 * jb + li) * ns + 16 * jb + 4 * q + lk];
      }
      const mfma_acc_t acc = mfma4(a, bw, mfma_acc_t{0.0, 0.0, 0.0, 0.0});
      double* Ct = S + (16 * it + lk) * ns + 16 * jb + li;
      Ct[0] = acc[0]; Ct[4 * ns] = acc[1]; Ct[8 * ns] = acc[2]; Ct[12 * ns] = acc[3];
    }
    __syncthreads();
    // trailing update: tile 0 = the next diagonal tile (first wavefront, which goes on to factor it), the others dealt
    // to the remaining wavefronts
    if (wave == 0) {
      trailing_tile(jb, jb + 1, jb + 1);
      diag(jb + 1);
    }
    if (NW == 1 || wave != 0) {
      int idx = 0;
      for (int it = jb + 1; it < NB; ++it)
        for (int ct = jb + 1; ct <= it; ++ct) {
          if (it == jb + 1 && ct == jb + 1) continue;
          if (NW == 1 || idx % (NW - 1) == wave - 1) trailing_tile(jb, it, ct);
          ++idx;
        }
      side(jb, wave);
    }
    __syncthreads();
  }
}

// wavefronts per SIMD the register budget is cut for: what LDS lets share a CU (two-round mode at n = 48, 64: three or
// four workgroups of NB wavefronts)
constexpr int reduced_min_waves(const int nb, const int nthr) { return nb >= 5 ? 1 : ((nthr == 64 * nb && nb >= 3) ? 3 : 4); }

template <int NB, int NTHR, bool LEVEL0, bool PAD>
__global__ __launch_bounds__(NTHR, reduced_min_waves(NB, NTHR)) void separator_reduced_mfma(Dims d, int l, const double* __restrict__ AB,
                                                                  const double* __restrict__ QR,
                                                                  const double* __restrict__ rhs, double* red,
                                                                  double* __restrict__ rec, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int n = 16 * NB, ns = reduced_s_pitch(n);
  constexpr int NW = NTHR / 64;
  constexpr int NT = NB * (NB + 1) / 2;          // tiles of the lower triangle of S-bar / DL / DR: all that is ever read
  constexpr int MAXS = (NT + NW - 1) / NW;       // S-bar tiles per wavefront
  constexpr int MAXT = (2 * NB + NW - 1) / NW;   // panel column tiles (of r_a, r_bb) per wavefront
  constexpr int PR = reduced_y_pitch(n);         // pitch of the panel halves in the push phase
  constexpr bool TWO = NW == NB;                 // two-round pushes on one coupling array (see above)
  static_assert(NW >= NB && NTHR >= n && n <= 128, "work distribution of the shared phases");
  // PAD: the block does not fill its tiles (nl < n rows / columns, or n + m not a multiple of 4). Everything in
  // global memory keeps the problem's own pitch nl; LDS holds the padded tiles: zero rows / columns of [A | B], r_a,
  // r_bb, DL, DR, unit diagonal of S-bar (so L, W are the identity there and the padding never reaches a result).
  const int nl = PAD ? d.n : n, nnl = nl * nl;
  const int w = d.w, wp = PAD ? (w + 3) / 4 * 4 : w, N = d.N, rows = d.rows;
  const int b = blockIdx.y;
  const int T = 2 << l, base = blockIdx.x * T, s = base + (1 << l) - 1;
  const bool hasA = base > 0, hasB = base + T < N, leftchild = (base & T) == 0, first = s == 0;
  double* dq = sm;           // 1 / [Q_s | R_s]  (state entries of knot 0: zero -- its state is fixed)
  double* zc = dq + wp;      // rhs(s).xu scaled likewise (state entries of knot 0: -x0)
  double* q1 = zc + (wp > n ? wp : n);  // 1 / Q_{s+1}  (zc doubles as the y of the z column: n entries)
  double* bz = q1 + n;       // b~
  double* S = bz + n;        // S-bar / L
  double* stage = S;         // [A_s | B_s], pitch P, until S-bar is formed
  double* Ra = S;            // r_a (pitch PR) once W is dead
  double* Rb = TWO ? S : S + (n * ns > n * PR ? n * ns : n * PR);  // Y of r_bb (pitch PR): written in phase B, or (two rounds) staged over that of r_a
  const int P = reduced_stage_pitch(wp);
  const int tid = threadIdx.x;
  // (the wavefront index as a scalar: everything that depends on it branches uniformly)
  // The roles of the wavefronts differ (the first factors the diagonal blocks, the second carries the rhs column, ..) and
  // wavefront k of every workgroup lands on SIMD k of its CU: the roles rotate with the workgroup, or one SIMD of
  // every CU carries all the diagonal blocks.
  const int lane = tid & 63;
  const int wave = (__builtin_amdgcn_readfirstlane(tid >> 6) + (int)((blockIdx.x + (blockIdx.x >> 3) + (blockIdx.x >> 6) + blockIdx.y) % NW)) % NW;
  const int li = lane & 15, lk = lane >> 4;
  SEG_INIT();

  const double* ab = AB + ((size_t)b * N + s) * nl * w;  // [A_s | B_s]
  const double* ab1 = ab + (size_t)nl * w;                // [A_{s+1} | B_{s+1}]  (s + 1 <= N - 1)
  const double* qr = QR + ((size_t)b * N + s) * w;
  const double* r0 = rhs + ((size_t)b * N + s) * rows;
  const double* myslot = reduced_slot(red, d, b, LEVEL0 ? 1 : s);  // (not read at level 0)
  double* slotA = reduced_slot(red, d, b, hasA ? base - 1 : 1);
  double* slotB = reduced_slot(red, d, b, hasB ? base + T - 1 : 1);
  double* myrec = rec + ((size_t)b * N + s) * (2 * (size_t)nnl + nl);
  const int ksteps = wp / 4;

  // ================================================================================================= phase A
  // ---- requests at kernel entry: [A_s | B_s] and the weights / rhs of knots s, s + 1
  const int kc = tid < w ? tid : w - 1, ic = tid < nl ? tid : nl - 1;
  const double qv = qr[kc], q1v = qr[w + ic], rxu = r0[nl + kc], rl0 = r0[ic], za = r0[rows + ic], zb = r0[rows + nl + ic];
  // DL + DR of this wavefront's S-bar tiles and gL + gR: consumed behind the leafS products
  double dlr[MAXS][8];
#pragma unroll
  for (int idx = 0; idx < MAXS; ++idx) {
    const int item = wave + idx * NW, itc = item < NT ? item : NT - 1, rt = tri_row(itc), ct = itc - rt * (rt + 1) / 2;
    const int jc = 16 * ct + li < nl ? 16 * ct + li : nl - 1;  // (clamped addresses; the padding is masked at the use)
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int i = 16 * rt + lk + 4 * gg, icl = i < nl ? i : nl - 1;
      dlr[idx][gg] = LEVEL0 ? 0.0 : myslot[(size_t)icl * nl + jc];
      dlr[idx][4 + gg] = LEVEL0 ? 0.0 : myslot[nnl + (size_t)icl * nl + jc];
    }
  }
  double gl = 0.0, gr = 0.0;
  if (!LEVEL0) { gl = myslot[4 * nnl + ic]; gr = myslot[4 * nnl + nl + ic]; }
  // ---- stage [A_s | B_s] (rows of pitch P; n = 64, w = 80: one round) and the diagonal weights / rhs of knots s, s + 1
  if constexpr (!PAD) {
    const int words = n * w / 2;  // 16-byte words of [A_s | B_s]
    const float inv_w = 1.0f / (float)w;
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
  } else {
    // every element of the padded block, zeros outside the nl x w data
    const int total = n * P;
    const float inv_p = 1.0f / (float)P;
    constexpr int SG = 8;
    for (int e0 = 0; e0 < total; e0 += SG * NTHR) {
      double st[SG];
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        const int e = e0 + tid + u * NTHR, ec = e < total ? e : total - 1;
        const int row = (int)(((float)ec + 0.5f) * inv_p), col = ec - row * P;
        st[u] = ab[(size_t)(row < nl ? row : nl - 1) * w + (col < w ? col : w - 1)];
      }
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        const int e = e0 + tid + u * NTHR, ec = e < total ? e : total - 1;
        const int row = (int)(((float)ec + 0.5f) * inv_p), col = ec - row * P;
        stage[ec] = (row < nl && col < w) ? st[u] : 0.0;
      }
    }
  }
  {
    const bool fx = first && kc < nl;
    const double inv = 1.0 / qv;
    if constexpr (!PAD) {
      dq[kc] = fx ? 0.0 : inv;
      zc[kc] = fx ? -rl0 : rxu * inv;
      q1[ic] = 1.0 / q1v;
    } else {
      // (the selects use the loaded values unconditionally; only the LDS stores sit under the predicates)
      const double dv = tid < w ? (fx ? 0.0 : inv) : 0.0, zv = tid < w ? (fx ? -rl0 : rxu * inv) : 0.0;
      const double qq = tid < nl ? 1.0 / q1v : 1.0;
      if (tid < wp) { dq[tid] = dv; zc[tid] = zv; }
      if (tid < n) q1[tid] = qq;
    }
    // the weights of knot s pass through exactly this separator (those of the last knot: Q through separator
    // N - 2, its R is not part of the problem): a non-positive one fails the Cholesky of Q_k / R_k in the reference
    // (src/nested_dissection.c:24-59) -- counted here, S-bar itself may well stay positive definite
    if ((tid < w && !(qv > 0.0)) || (s == N - 2 && tid < nl && !(q1v > 0.0))) flag_failure(info, d, b);
  }
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
      const int item = wave + idx * NW, itc = item < NT ? item : NT - 1, rt = tri_row(itc), ct = itc - rt * (rt + 1) / 2;
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
  // b~ = [A_s | B_s] zc - z(s+1).lambda - z(s+1).x / Q_{s+1} - gL - gR: eight lanes per row (lane = (row, k mod 8):
  // at most two-way bank conflicts where a row per lane has eight-way ones), rows dealt to the wavefronts
  for (int i0 = 8 * wave; i0 < n; i0 += 8 * NW) {
    const int i = i0 + (lane >> 3);
    const double* arow = stage + i * P + (lane & 7);
    double acc = 0.0;
    for (int k0 = 0; k0 < w; k0 += 32) {
      double av[4], zv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 8 * u + (lane & 7), kk = k < w ? k : w - 1;
        av[u] = arow[kk - (lane & 7)];
        zv[u] = k < w ? zc[kk] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = fma(av[u], zv[u], acc);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if ((lane & 7) == 0) bz[i] = acc;  // (bz is not read before the barrier below)
  }
  __syncthreads();
  SEG(51);
  {
    // (an unconditional use of the early requests: under the predicate alone the compiler would sink the loads there)
    const double corr = fma(zb, q1[ic], za) + (gl + gr);
    if (tid < nl) bz[tid] -= corr;
  }
#pragma unroll
  for (int idx = 0; idx < MAXS; ++idx) {
    const int item = wave + idx * NW;
    if (item < NT) {
      const int rt = tri_row(item), ct = item - rt * (rt + 1) / 2;
      double* dst = S + (16 * rt + lk) * ns + 16 * ct + li;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const int i = 16 * rt + lk + 4 * gg, j = 16 * ct + li;
        const double dd = (!PAD || (i < nl && j < nl)) ? dlr[idx][gg] + dlr[idx][4 + gg] : 0.0;
        dst[4 * gg * ns] = sacc[idx][gg] + (i == j ? q1[i] : 0.0) - dd;
      }
    }
  }
  __syncthreads();
  SEG(52);

  // ---- blocked Cholesky. Beside it, as soon as a block row of the factor is final: the second wavefront takes the
  //      right-hand-side column through the forward substitution (y~ = L^-1 b~, in place over b~), the others store the
  //      block row into the compact record -- L with the inverses of its diagonal blocks in their place, lower triangle
  //      packed with the problem's own size (entry (i, k), k <= i, at i (i + 1) / 2 + k), and y~, instead of
  //      f_a | f_bb | z_sep: a quarter of the bytes, and X = L^-T Y is never formed. The back-substitution
  //      (backsub_multipliers_compact, level 0: backsub_level0_states_generic) forms f_a y_A + f_bb y_B =
  //      S-bar^-1 (r_a y_A + r_bb y_B) from the couplings, which stay where this kernel reads them (slot / data).
  auto At = [&](const int i, const int k) -> double { return S[i * ns + k]; };
  reduced_cholesky<NB, NW>(S, ns, lane, wave, info, d, b, [&](const int jb, const int wv) {
    constexpr int YW = NW > 1 ? 1 : 0;                     // the wavefront of y~
    constexpr int SW0 = NW >= 3 ? 2 : YW, NSW = NW - SW0;  // the wavefronts that store
    if (wv == YW) {
      tri_forward_block(jb, n, bz, zc, lane, At, At);
      if (jb == NB - 1)
        for (int i = lane; i < nl; i += 64) myrec[2 * nnl + i] = bz[i];
    }
    if (wv >= SW0) {
      constexpr int KC = (n + 63) / 64;  // lanes along a row of L: one or two entries each
      for (int r0 = wv - SW0; r0 < 16; r0 += 4 * NSW) {
        double v[4][KC];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = 16 * jb + r0 + u * NSW, ic = i < n ? i : n - 1;
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) v[u][kc] = S[ic * ns + (lane + 64 * kc <= ic ? lane + 64 * kc : ic)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + u * NSW, i = 16 * jb + r;
#pragma unroll
          for (int kc = 0; kc < KC; ++kc)
            if (r < 16 && i < nl && lane + 64 * kc <= i) myrec[(size_t)i * (i + 1) / 2 + lane + 64 * kc] = v[u][kc];
        }
      }
    }
  });
  SEG(53);
  // the column tile(s) of [r_a | r_bb] this wavefront will solve: requested here, consumed behind the last block of y~
  // (not before the Cholesky: its diagonal-block wavefront needs the registers, and a spilled load waits)
  double rfk[MAXT][NB][4];  // element (16 kb + 4 q + lk, 16 c + li) of r_a / r_bb (raw operand until phase B)
#pragma unroll
  for (int m = 0; m < MAXT; ++m) {
    const int gt = wave + m * NW, gc = gt < 2 * NB ? gt : 2 * NB - 1, c = gc < NB ? gc : gc - NB;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 16 * kb + 4 * q + lk, k = k0 < nl ? k0 : nl - 1, j = 16 * c + li < nl ? 16 * c + li : nl - 1;
        // (unconditional on clamped indices: a coupling block that does not exist is a valid address with
        //  arbitrary content; it and the padding are masked in phase B.) level 0: r_bb(k, j) = -A_{s+1}(j, k) / Q_{s+1}(k)
        if (gc < NB) rfk[m][kb][q] = LEVEL0 ? ab[(size_t)k * w + j] : myslot[2 * nnl + k * nl + j];
        else rfk[m][kb][q] = LEVEL0 ? ab1[(size_t)j * w + k] : myslot[3 * nnl + k * nl + j];
      }
  }
  SEG(54);
  // ================================================================================================= phase B
  // Column tile gt of the panel: [0, NB) columns of r_a, [NB, 2 NB) of r_bb. Y = L^-1 R by block forward
  // substitution in the accumulators: yk[m][kb][q] = Y(16 kb + 4 q + lk, 16 c + li) -- an accumulator tile is at once
  // the B operand of k-steps 4 kb .. 4 kb + 3 (and, transposed, the A operand) of the products that follow, so
  // nothing goes through LDS and nobody waits for anybody. Right-looking: block kb of Y, once final, goes into the
  // accumulators of all blocks below it.
  mfma_acc_t yk[MAXT][NB];
#pragma unroll
  for (int m = 0; m < MAXT; ++m) {
    const int gt = wave + m * NW;
    double (&rf)[NB][4] = rfk[m];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) yk[m][kb] = mfma_acc_t{0.0, 0.0, 0.0, 0.0};
    if (gt >= 2 * NB) continue;
    const int c = gt < NB ? gt : gt - NB;
    // the accumulators start from -R (r_a = -CA, r_bb = -CB: the raw operand; level 0: scaled by the weights)
    if (gt < NB) {
      const double dj = dq[16 * c + li];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool in = hasA && (!PAD || (16 * kb + 4 * q + lk < nl && 16 * c + li < nl));
          yk[m][kb][q] = !in ? 0.0 : LEVEL0 ? rf[kb][q] * dj : rf[kb][q];
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool in = hasB && (!PAD || (16 * kb + 4 * q + lk < nl && 16 * c + li < nl));
          yk[m][kb][q] = !in ? 0.0 : LEVEL0 ? rf[kb][q] * q1[16 * kb + 4 * q + lk] : rf[kb][q];
        }
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      // Y_kb = D_kb^-1 (R_kb - sum_{j < kb} L_kb,j Y_j) = (-D_kb^-1) (accumulator)
      double a[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = -S[(16 * kb + li) * ns + 16 * kb + 4 * q + lk];
      mfma_acc_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q) y = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], yk[m][kb][q], y, 0, 0, 0);
      yk[m][kb] = y;
#pragma unroll
      for (int it = kb + 1; it < NB; ++it) {
        double al[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) al[q] = S[(16 * it + li) * ns + 16 * kb + 4 * q + lk];
#pragma unroll
        for (int q = 0; q < 4; ++q) yk[m][it] = __builtin_amdgcn_mfma_f64_16x16x4f64(al[q], y[q], yk[m][it], 0, 0, 0);
      }
    }
    if constexpr (!TWO) {
      if (gt >= NB) {  // Y of r_bb: operand of the push phase, into its own array
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
          for (int q = 0; q < 4; ++q) Rb[(16 * kb + 4 * q + lk) * PR + 16 * c + li] = yk[m][kb][q];
      }
    }
  }
  SEG(55);
  __syncthreads();  // S-bar / L are dead
  SEG(56);

  // ================================================================================================= phase C
  // Y of r_a (two rounds: later Y of r_bb) from the registers into the coupling array over the dead S-bar / L
  auto stage_y = [&](const bool a_side) {
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      const int gt = wave + m * NW;
      if (gt < 2 * NB && (gt < NB) == a_side) {
        const int c = gt < NB ? gt : gt - NB;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
          for (int q = 0; q < 4; ++q) S[(16 * kb + 4 * q + lk) * PR + 16 * c + li] = yk[m][kb][q];
      }
    }
  };
  stage_y(true);
  __syncthreads();
  SEG(58);

  // ================================================================================================= phase D
  // tile product of the push phase: acc (+)= R(:, 16 rt ..)' X  (xa: X' R(:, 16 rt ..)), R from LDS, X from registers
  const mfma_acc_t zero4 = {0.0, 0.0, 0.0, 0.0};
  // blk: n x n block of a slot (row pitch nl), the tile at rows 16 orow .., columns 16 ocol ..; acc: what it starts from
  auto push_tile = [&](const double* Rl, const int rt, const mfma_acc_t (&x)[NB], const bool xa, double* blk,
                       const int orow, const int ocol, mfma_acc_t acc) {
    const double* rl = Rl + lk * PR + 16 * rt + li;
    constexpr int QH = (n / 4) % 8 == 0 ? 8 : ((n / 4) % 6 == 0 ? 6 : 4);  // operand fragments of 8 (6, 4) k-steps at a time
    static_assert((n / 4) % QH == 0, "whole rounds");
#pragma unroll
    for (int q0 = 0; q0 < n / 4; q0 += QH) {
      double rv[QH];
#pragma unroll
      for (int q = 0; q < QH; ++q) rv[q] = rl[4 * (q0 + q) * PR];
      if (xa) {
#pragma unroll
        for (int q = 0; q < QH; ++q)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[(q0 + q) / 4][(q0 + q) % 4], rv[q], acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int q = 0; q < QH; ++q)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[q], x[(q0 + q) / 4][(q0 + q) % 4], acc, 0, 0, 0);
      }
    }
    if (!PAD || 16 * ocol + li < nl) {
      double* dst = blk + (size_t)(16 * orow + lk) * nl + 16 * ocol + li;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg)
        if (!PAD || 16 * orow + lk + 4 * gg < nl) dst[(size_t)4 * gg * nl] = acc[gg];
    }
  };
  // Everything that multiplies with r_a (with_ra) or with r_bb. DR, DL and S-bar are symmetric: only tiles on and
  // below the diagonal. The coupling tile (a-tile i, bb-tile j) goes to the wavefront of f_a's tile i when j <= i
  // (as f_a' r_bb = r_a' S-bar^-1 r_bb) and to the one of f_bb's tile j when i < j (as r_a' f_bb).
  auto pushes = [&](const bool with_ra) {
    const double* Rl = with_ra ? Ra : Rb;
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      const int gt = wave + m * NW;
      if (gt >= 2 * NB) continue;
      const int c = gt < NB ? gt : gt - NB;
      const bool atile = gt < NB;
      if (atile ? !hasA : !hasB) continue;
      if (atile == with_ra) {
        // DR[A](lower tiles, tile column c) += r_a' f_a(:, c)   /   DL[B](..) += r_bb' f_bb(:, c); the tiles they start
        // from are requested first
        double* pblk = atile ? slotA + nnl : slotB;
        const int jc = 16 * c + li < nl ? 16 * c + li : nl - 1;
        // (tiles above the diagonal are neither kept nor read; padding: never stored)
        auto start_tile = [&](const int rt) -> mfma_acc_t {
          mfma_acc_t a0 = {0.0, 0.0, 0.0, 0.0};
          if (!LEVEL0 && rt >= c) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
              const int i = 16 * rt + lk + 4 * gg;
              a0[gg] = pblk[(size_t)(i < nl ? i : nl - 1) * nl + jc];
            }
          }
          return a0;
        };
        if constexpr (TWO) {  // (registers are short with two solved tiles per wavefront: tile by tile)
#pragma unroll
          for (int rt = 0; rt < NB; ++rt)
            if (rt >= c) push_tile(Rl, rt, yk[m], false, pblk, rt, c, start_tile(rt));
        } else {
          mfma_acc_t pacc[NB];
#pragma unroll
          for (int rt = 0; rt < NB; ++rt) pacc[rt] = start_tile(rt);
#pragma unroll
          for (int rt = 0; rt < NB; ++rt)
            if (rt >= c) push_tile(Rl, rt, yk[m], false, pblk, rt, c, pacc[rt]);
        }
      } else if (hasA && hasB) {
        if (atile) {  // with r_bb: coupling tiles (c, j <= c) as f_a' r_bb
          for (int j = 0; j <= c; ++j) {
            if (leftchild) push_tile(Rl, j, yk[m], false, slotB + 2 * nnl, j, c, zero4);  // CA[B] = (f_a' r_bb)': rows bb-tile j
            else push_tile(Rl, j, yk[m], true, slotA + 3 * nnl, c, j, zero4);           // CB[A] = f_a' r_bb: rows a-tile c
          }
        } else {      // with r_a: coupling tiles (i < c, c) as r_a' f_bb
          for (int i = 0; i < c; ++i) {
            if (leftchild) push_tile(Rl, i, yk[m], true, slotB + 2 * nnl, c, i, zero4);   // CA[B] = f_bb' r_a: rows bb-tile c
            else push_tile(Rl, i, yk[m], false, slotA + 3 * nnl, i, c, zero4);          // CB[A] = r_a' f_bb: rows a-tile i
          }
        }
      }
    }
    // vector pushes gR[A] += Y_a' y~ (first wavefront) / gL[B] += Y_b' y~ (last): a column of Y per lane
    if (with_ra ? (wave == 0 && hasA) : (wave == NW - 1 && hasB)) {
      double* dst = with_ra ? slotA + 4 * nnl + nl : slotB + 4 * nnl;
      for (int j0 = 0; j0 < nl; j0 += 64) {
        const int j = j0 + lane < nl ? j0 + lane : nl - 1;
        double acc = LEVEL0 ? 0.0 : dst[j];
#pragma unroll 1
        for (int k0 = 0; k0 < n; k0 += 8) {  // (not unrolled: the solved tiles are still live for the second round)
          double rv[8], zv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { rv[u] = Rl[(k0 + u) * PR + j]; zv[u] = bz[k0 + u]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = fma(rv[u], zv[u], acc);
        }
        if (j0 + lane < nl) dst[j] = acc;
      }
    }
  };
  pushes(true);
  if constexpr (TWO) {
    __syncthreads();  // r_a has been read
    stage_y(false);
    __syncthreads();
  }
  pushes(false);
  SEG(59);
}


// ------------------------------------------------------------------------------------- rhs-only re-solve
// New q, r, d, x0 against the factorisation a separator_reduced_mfma sweep left behind (the solver
// keeps it when NDLQR_FLAG_KEEP_RECORDS is set): per level, vector work only --
//     b~ = leafb - gL - gR,   y~ = L^-1 b~  -> record,   z_sep = L^-T y~,   gR[A] (+)= r_a' z_sep,   gL[B] (+)= r_bb' z_sep
// with L (packed, the inverses of its diagonal blocks in their place) from the separator's compact record, r_a = -CA,
// r_bb = -CB from the slots (level 0: from the problem data). The back-substitution then runs as after a full solve.
// Runtime-sized; np = padded block size of the factorisation.
//   grid (N >> (l+1), batch), block 256, dynamic LDS = 2 (n + m) + n + 3 np + 256 + n (n + 1) / 2 doubles.
static __global__ __launch_bounds__(256) void rhs_reduced_generic(Dims d, int l, int np, const double* __restrict__ AB,
                                                                  const double* __restrict__ QR,
                                                                  const double* __restrict__ rhs, double* red,
                                                                  double* __restrict__ rec) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int nl = d.n, nnl = nl * nl, w = d.w, N = d.N, rows = d.rows;
  const int b = blockIdx.y;
  const int T = 2 << l, base = blockIdx.x * T, s = base + (1 << l) - 1;
  const bool hasA = base > 0, hasB = base + T < N, first = s == 0, level0 = l == 0;
  double* dq = sm;        // 1 / [Q_s | R_s] (state entries of knot 0: zero)
  double* zc = dq + w;    // rhs(s).xu scaled likewise (state entries of knot 0: -x0)
  double* q1 = zc + w;    // 1 / Q_{s+1}
  double* bz = q1 + nl;   // b~
  double* yv = bz + np;   // scratch of the block substitutions
  double* zs = yv + np;   // z_sep
  double* part = zs + np; // partial column sums: 256 / np segments of the summation index per column
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* ab = AB + ((size_t)b * N + s) * nl * w;
  const double* ab1 = ab + (size_t)nl * w;
  const double* qr = QR + ((size_t)b * N + s) * w;
  const double* r0 = rhs + ((size_t)b * N + s) * rows;
  const double* myslot = reduced_slot(red, d, b, level0 ? 1 : s);
  double* slotA = reduced_slot(red, d, b, hasA ? base - 1 : 1);
  double* slotB = reduced_slot(red, d, b, hasB ? base + T - 1 : 1);
  double* myrec = rec + ((size_t)b * N + s) * (2 * (size_t)nnl + nl);
  double* Lp = part + 256;  // the separator's factor from its compact record: packed lower triangle, staged
  auto P = [&](const int r, const int c) -> double { return Lp[r * (r + 1) / 2 + c]; };
  auto wave_sum = [](double v) -> double {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
  {  // the factor: every load in flight before the first LDS store
    const int WF = nl * (nl + 1) / 2;
    for (int e0 = 0; e0 < WF; e0 += 8 * 256) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + tid + 256 * u; t[u] = myrec[e < WF ? e : WF - 1]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + tid + 256 * u; if (e < WF) Lp[e] = t[u]; }
    }
  }

  for (int k = tid; k < w; k += 256) {
    const bool fx = first && k < nl;
    const double inv = 1.0 / qr[k];
    dq[k] = fx ? 0.0 : inv;
    zc[k] = fx ? -r0[k] : r0[nl + k] * inv;
  }
  for (int i = tid; i < nl; i += 256) q1[i] = 1.0 / qr[w + i];
  __syncthreads();
  // b~: a row per wavefront and round, lanes along the row
  for (int i = wave; i < np; i += 4) {
    double acc = 0.0;
    if (i < nl)
      for (int k = lane; k < w; k += 64) acc = fma(ab[(size_t)i * w + k], zc[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      double v = 0.0;
      if (i < nl) {
        v = acc - fma(r0[rows + nl + i], q1[i], r0[rows + i]);
        if (!level0) v -= myslot[4 * nnl + i] + myslot[4 * nnl + nl + i];
      }
      bz[i] = v;
    }
  }
  __syncthreads();
  // y~ = L^-1 b~ -> record, and z_sep = L^-T y~ for the pushes (block substitutions, first wavefront)
  if (wave == 0) {
    const int nblk = (nl + 15) >> 4;
    for (int ib = 0; ib < nblk; ++ib) tri_forward_block(ib, nl, bz, yv, lane, P, P);
    for (int i = lane; i < np; i += 64) {
      if (i < nl) myrec[2 * nnl + i] = bz[i];
      zs[i] = i < nl ? bz[i] : 0.0;
    }
    wave_lds_order();
    for (int ib = nblk - 1; ib >= 0; --ib) tri_backward_block(ib, nl, zs, yv, lane, P, P);
  }
  __syncthreads();
  // column sums sum_k f(k, j): thread (j, seg) takes k = seg, seg + nseg, ..; consecutive threads read consecutive
  // words of a row of the operand
  const int nseg = 256 / np, cj = tid % np, cseg = tid / np;
  auto column_sums = [&](auto f, const int kend) -> double {  // returns the sum of column cj to the threads of segment 0
    double acc = 0.0;
    if (cseg < nseg)
      for (int k = cseg; k < kend; k += nseg) acc += f(k, cj);
    if (cseg < nseg) part[cseg * np + cj] = acc;
    __syncthreads();
    double tot = 0.0;
    if (cseg == 0)
      for (int g = 0; g < nseg; ++g) tot += part[g * np + cj];
    __syncthreads();
    return tot;
  };
  // gR[A] (+)= r_a' z_sep: a column of r_a per thread
  if (hasA) {  // (uniform)
    double* dst = slotA + 4 * nnl + nl;
    const double g = column_sums([&](const int k, const int j) {
      if (j >= nl) return 0.0;
      return (level0 ? -ab[(size_t)k * w + j] * dq[j] : -myslot[2 * nnl + k * nl + j]) * zs[k];
    }, nl);
    if (cseg == 0 && cj < nl) dst[cj] = level0 ? g : dst[cj] + g;
  }
  // gL[B] (+)= r_bb' z_sep; level 0: r_bb(k, j) = -A_{s+1}(j, k) / Q_{s+1}(k) -- a row of A_{s+1} per wavefront and round
  if (hasB) {
    double* dst = slotB + 4 * nnl;
    if (level0) {
      for (int j = wave; j < nl; j += 4) {
        double acc = 0.0;
        for (int k = lane; k < nl; k += 64) acc = fma(-q1[k] * ab1[(size_t)j * w + k], zs[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) dst[j] = acc;
      }
    } else {
      const double g = column_sums([&](const int k, const int j) {
        return j < nl ? -myslot[3 * nnl + k * nl + j] * zs[k] : 0.0;
      }, nl);
      if (cseg == 0 && cj < nl) dst[cj] += g;
    }
  }
}

// ------------------------------------------------------------------------------------- back-substitution, levels >= 1
// Multipliers of the separators of level l >= 1 from the compact records, top-down (one launch per level):
//     y_s = S-bar^-1 (b~ - r_a y_A - r_bb y_B) = L^-T (y~ + L^-1 (CA y_A + CB y_B))
// with CA = -r_a, CB = -r_bb where the factorisation read them (the separator's slot: nothing writes it after its
// elimination), L and y~ from the record; y_A, y_B are final (higher levels) in the lambda rows of knots A + 1, B + 1.
// One workgroup per separator: the rows of CA | CB are dealt to the wavefronts eight at a time (lanes along a row:
// coalesced, sixteen loads in flight per lane), the two block substitutions run on the first wavefront.
//   grid (N >> (l+1), batch), block 64 / 128 / 256 (by block size: a workgroup of four wavefronts per 16 x 16 separator left
//   three of them idle and the CU a quarter full), dynamic LDS = n (n + 1) / 2 + 2 n + 16 doubles. (A step that wants a knot range
//   alone runs the separators above it: Dims::xoff = the first one's index on the level, a shorter grid.)
static __global__ __launch_bounds__(256) void backsub_multipliers_compact(Dims d, int l, const double* __restrict__ red,
                                                                          const double* __restrict__ recs, double* z) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = d.n, nn = n * n, rows = d.rows, N = d.N, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
  const int T = 2 << l, base = (blockIdx.x + d.xoff) * T, s = base + (1 << l) - 1;
  const bool hasA = base > 0, hasB = base + T < N;
  double* Lp = sm;                     // L / inverses of its diagonal blocks, packed lower triangle
  double* tv = Lp + n * (n + 1) / 2;   // CA y_A + CB y_B, then the substitutions
  double* yt = tv + n;                 // y~
  double* tmp = yt + n;                // scratch of the substitutions (16)
  const double* rc = recs + ((size_t)b * N + s) * (2 * (size_t)nn + n);
  const double* slot = red + ((size_t)b * (N >> 1) + (s >> 1)) * (4 * (size_t)nn + 2 * n);
  const double* CA = slot + 2 * (size_t)nn;
  const double* CB = slot + 3 * (size_t)nn;
  const double* yA = z + ((size_t)b * N + base) * rows;      // y_A lives in the lambda rows of knot A + 1 = base
  const double* yB = z + ((size_t)b * N + base + T) * rows;
  double* out = z + ((size_t)b * N + s + 1) * rows;
  {  // the factor: every load in flight before the first LDS store
    const int np = n * (n + 1) / 2;
    for (int e0 = 0; e0 < np; e0 += 8 * nthr) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + tid + nthr * u; t[u] = rc[e < np ? e : np - 1]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int e = e0 + tid + nthr * u; if (e < np) Lp[e] = t[u]; }
    }
    for (int i = tid; i < n; i += nthr) yt[i] = rc[2 * (size_t)nn + i];
  }
  for (int r0 = 8 * wave; r0 < n; r0 += 8 * nwave) {
    double part[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) part[u] = 0.0;
    for (int c = lane; c < n; c += 64) {
      double fa[8], fb[8], ya = 0.0, yb = 0.0;
      if (hasA) {  // uniform
        ya = yA[c];
#pragma unroll
        for (int u = 0; u < 8; ++u) fa[u] = CA[(size_t)(r0 + u < n ? r0 + u : n - 1) * n + c];
      }
      if (hasB) {
        yb = yB[c];
#pragma unroll
        for (int u = 0; u < 8; ++u) fb[u] = CB[(size_t)(r0 + u < n ? r0 + u : n - 1) * n + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (hasA) part[u] = fma(fa[u], ya, part[u]);
        if (hasB) part[u] = fma(fb[u], yb, part[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      double p = part[u];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
      if (lane == 0 && r0 + u < n) tv[r0 + u] = p;
    }
  }
  __syncthreads();
  if (wave == 0) {
    auto P = [&](const int i, const int k) -> double { return Lp[i * (i + 1) / 2 + k]; };
    const int nblk = (n + 15) >> 4;
    for (int ib = 0; ib < nblk; ++ib) tri_forward_block(ib, n, tv, tmp, lane, P, P);
    for (int i = lane; i < n; i += 64) tv[i] += yt[i];
    wave_lds_order();
    for (int ib = nblk - 1; ib >= 0; --ib) tri_backward_block(ib, n, tv, tmp, lane, P, P);
    for (int i = lane; i < n; i += 64) out[i] = tv[i];
  }
}

// ------------------------------------------------------------------------------------- back-substitution, level 0
// The last step of the back-substitution of the separator-only schedule above, one workgroup per level-0
// separator s = 2 j, i.e. per pair of knots (s, s + 1): its multiplier from the compact record
//     y_s = L^-T (y~ - L^-1 (r_a y_{s-1} + r_bb y_{s+1})),   r_a = -A_s Q_s^-1,  r_bb = -Q_{s+1}^-1 A_{s+1}'
// (the multipliers next to it are final: levels >= 1 ran before, backsub_multipliers_compact), then the states
// and inputs of both knots (the arithmetic of backsub_states_generic) -- [A | B] of the two knots comes from HBM
// once for both, and level 0's f_a | f_bb never exist in memory.
//   grid (N / 2, batch), block 64 / 128 / 256 (by block size), dynamic LDS = n (n + 1) / 2 + 5 n + 4 (n + m) +
//   2 (2 n + m) + block doubles.
static __global__ __launch_bounds__(256) void backsub_level0_states_generic(Dims d, const double* __restrict__ AB,
                                                                            const double* __restrict__ QR,
                                                                            const double* __restrict__ rhs,
                                                                            const double* __restrict__ rec,
                                                                            double* z) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = d.n, w = d.w, N = d.N, rows = d.rows, nn = n * n;
  const int b = blockIdx.y, s = 2 * (blockIdx.x + d.xoff);
  const bool hasA = s > 0, hasB = s + 2 < N;
  double* Wp = sm;                      // L (the inverses of its diagonal blocks in their place), packed lower triangle
  double* yA = Wp + n * (n + 1) / 2;    // y_{s-1}, then y_{s-1} / Q_s
  double* yB = yA + n;                  // y_{s+1}
  double* ys = yB + n;                  // y_s
  double* tv = ys + n;                  // r_a y_A + r_bb y_B, then the substitutions
  double* zs = tv + n;                  // y~ of the record
  double* d0 = zs + n;                  // [A_s | B_s]' y_s
  double* d1 = d0 + w;                  // [A_{s+1} | B_{s+1}]' y_{s+1}
  double* qv = d1 + w;                  // [Q | R] of knots s, s + 1
  double* rv = qv + 2 * w;              // raw right-hand sides of knots s, s + 1
  double* part = rv + 2 * rows;         // partial column sums (one per thread)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
  const double* ab = AB + ((size_t)b * N + s) * n * w;
  const double* ab1 = ab + (size_t)n * w;
  const double* qr = QR + ((size_t)b * N + s) * w;
  const double* r0 = rhs + ((size_t)b * N + s) * rows;
  const double* myrec = rec + ((size_t)b * N + s) * (2 * (size_t)nn + n);
  double* zk = z + ((size_t)b * N + s) * rows;  // knot s; y_{s-1} lives in its lambda rows, y_s in those of knot s + 1
  auto wave_sum = [](double v) -> double {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
  // out[c] = sum_j m[j * w + c] y[j], c < w (a block transposed times a vector): thread (c, seg) sums the rows
  // j = seg, seg + nseg, .. -- consecutive threads read consecutive words of a row of the block
  // (m: first column of the range, nc columns, row pitch w)
  auto block_t_times = [&](const double* m, const int nc, const double* y, double* out) {
    const int nseg = 2 * nc <= nthr ? nthr / nc : 1;
    if (nseg > 1) {
      const int c = tid % nc, seg = tid / nc;
      double acc = 0.0;
      if (seg < nseg) {
        for (int j0 = seg; j0 < n; j0 += 8 * nseg) {  // eight loads in flight
          double mv[8], yv8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * nseg, jc = j < n ? j : n - 1;
            mv[u] = m[(size_t)jc * w + c];
            yv8[u] = j < n ? y[jc] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = fma(mv[u], yv8[u], acc);
        }
        part[tid] = acc;
      }
      __syncthreads();
      if (seg == 0) {
        double tot = 0.0;
        for (int g = 0; g < nseg; ++g) tot += part[g * nc + c];
        out[c] = tot;
      }
      __syncthreads();
    } else {
      for (int c = tid; c < nc; c += nthr) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc = fma(m[(size_t)j * w + c], y[j], acc);
        out[c] = acc;
      }
      __syncthreads();
    }
  };

  // (requested first, consumed behind the pass over [A_{s+1} | B_{s+1}]: see the product with A_s below)
  const bool wide = n > 64;  // (uniform) more than 64 states: A_s does not fit the registers this way, it is read twice
  double areg[16];
  {
    const int jc = lane < n ? lane : n - 1;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = wave + u * nwave;
      areg[u] = wide ? 0.0 : ab[(size_t)(i < n ? i : n - 1) * w + jc];
    }
  }
  for (int e = tid; e < n * (n + 1) / 2; e += nthr) Wp[e] = myrec[e];
  for (int i = tid; i < n; i += nthr) {
    yA[i] = hasA ? zk[i] : 0.0;
    yB[i] = hasB ? zk[2 * rows + i] : 0.0;
    zs[i] = myrec[2 * nn + i];
  }
  for (int e = tid; e < 2 * w; e += nthr) qv[e] = qr[e];
  for (int e = tid; e < 2 * rows; e += nthr) rv[e] = r0[e];
  __syncthreads();
  block_t_times(ab1, w, yB, d1);  // [A_{s+1} | B_{s+1}]' y_{s+1}: enters t, x_{s+1} and u_{s+1}
  // t = r_a y_A + r_bb y_B = -A_s (y_A / Q_s) - (A_{s+1}' y_{s+1}) / Q_{s+1}. A_s stays in registers for the second
  // product with it further down (A_s' y_s), up to 64 states: wavefront wv holds the rows wv, wv + nwave, .. (at most
  // sixteen: n <= 64 with four wavefronts, <= 32 with two, <= 16 with one), lane = column -- every row is one coalesced request, all of
  // them in flight together, and the block comes from HBM once.
  if (!wide) {
    const double yq = (hasA && lane < n) ? yA[lane] / qv[lane] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = wave + u * nwave;
      const double a = wave_sum(areg[u] * yq);
      if (lane == 0 && i < n) tv[i] = -a - (hasB ? d1[i] / qv[w + i] : 0.0);
    }
  } else {
    for (int i0 = 4 * wave; i0 < n; i0 += 4 * nwave) {  // four rows of A_s per wavefront and round, lanes along the row
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      if (hasA) {
        for (int j = lane; j < n; j += 64) {
          const double yq = yA[j] / qv[j];
          double av[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) av[u] = ab[(size_t)(i0 + u < n ? i0 + u : n - 1) * w + j];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] = fma(av[u], yq, acc[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double a = wave_sum(acc[u]);
        const int i = i0 + u;
        if (lane == 0 && i < n) tv[i] = -a - (hasB ? d1[i] / qv[w + i] : 0.0);
      }
    }
  }
  __syncthreads();
  // y_s = L^-T (y~ - L^-1 t): two block substitutions on the packed factor (first wavefront)
  if (wave == 0) {
    auto P = [&](const int i, const int k) -> double { return Wp[i * (i + 1) / 2 + k]; };
    const int nblk = (n + 15) >> 4;
    for (int ib = 0; ib < nblk; ++ib) tri_forward_block(ib, n, tv, part, lane, P, P);
    for (int i = lane; i < n; i += 64) tv[i] = zs[i] - tv[i];
    wave_lds_order();
    for (int ib = nblk - 1; ib >= 0; --ib) tri_backward_block(ib, n, tv, part, lane, P, P);
  }
  __syncthreads();
  if (tid < n) {
    const double yi = tv[tid];
    ys[tid] = yi;
    zk[rows + tid] = yi;  // lambda rows of knot s + 1
  }
  __syncthreads();
  // [A_s | B_s]' y_s: the state columns from the registers (partial sums per wavefront, then across them), the input
  // columns from memory (their first and only pass)
  if (!wide) {
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = wave + u * nwave;
      acc = fma(areg[u], i < n ? ys[i] : 0.0, acc);
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < n) {
      double tot = 0.0;
      for (int g = 0; g < nwave; ++g) tot += part[64 * g + tid];
      d0[tid] = tot;
    }
    __syncthreads();
    if (w > n) block_t_times(ab + n, w - n, ys, d0 + n);
  } else {
    block_t_times(ab, w, ys, d0);
  }
  // states and inputs of knots s and s + 1 (the arithmetic of backsub_states_generic); thread -> (knot, row)
  for (int e = tid; e < 2 * rows; e += nthr) {
    const int kk = e / rows, r = e - kk * rows, k = s + kk;
    if (r < n && k > 0) continue;  // lambda rows of knots >= 1 are the multipliers already
    const double* ykm = kk ? ys : yA;     // y_{k-1}
    const double* q = qv + kk * w;
    const double* rr = rv + kk * rows;
    const int col = r < n ? r : r - n;    // column of [A_k | B_k] this row meets
    double dot = 0.0;
    if (k < N - 1 && !(k == 0 && r >= n && r < 2 * n)) dot = (kk ? d1 : d0)[col];
    double out;
    if (r < n) out = fma(-q[r], rr[r], -rr[n + r]) + dot;                                 // knot 0: Q x0 + q + A_0' y_0
    else if (r < 2 * n) out = (k == 0) ? -rr[r - n] : (rr[r] - dot + ykm[r - n]) / q[r - n];  // x_k
    else out = (k == N - 1) ? rr[r] : (rr[r] - dot) / q[r - n];                            // u_k
    zk[(size_t)kk * rows + r] = out;
  }
}

}  // namespace ndlqr
