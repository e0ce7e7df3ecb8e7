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
#include "kernels_dpp.hpp"
#include "kernels_leaf.hpp"

namespace ndlqr {

// ------------------------------------------------------------------------------------- separators
// grid (N / 2^(l+1), batch), block 256, dynamic LDS = (n (n+1) + n (2n+1)) doubles -- or, for blocks whose S-bar and
// panel do not fit the 160 KB of LDS (beyond ~80 states on this path), `scratch` != nullptr: the same two arrays per
// workgroup in global memory, [batch][N / 2^(l+1)][n (n+1) + n (2n+1)] doubles (they stay in the L2 of the workgroup's
// XCD; __syncthreads orders the workgroup's global accesses like its LDS accesses). Same code, same arithmetic: every
// block size the device memory holds is solvable in every mode, slowly.
// For the level-l separator s of each subtree: S-bar, the two outer right-hand sides f_a, f_bb and
// the rhs vector; Cholesky; solves; results stored in the lambda rows of knot s+1.
// LDS: S-bar / L with rows padded to n+1 (lane i walks row i: no bank conflicts), and ONE panel
// X[n][2n+1] = [f_a | f_bb | z_sep]. Cholesky (right-looking) and both substitutions are
// "scale row/column j, then rank-1 update of the rest" with a wavefront per row and a lane per
// column -- the whole workgroup works on every pivot, no integer division in the loops. Every
// element receives its updates in the reference's order (k resp. j ascending; descending in the
// transposed sweep), so the results equal the left-looking / column-by-column reference loops.
// (Fast mode with n a multiple of 16 and n+m of 4 runs separator_mfma of kernels_mfma.hpp instead: the
// same separator on v_mfma_f64_16x16x4_f64, blocked by 16 columns.)
// Fast path of the blocked separator: lower Cholesky of one 16x16 diagonal block AND the inverse
// of that factor by ONE wavefront in registers (lanes 0..15 own a row, then a column; row broadcasts
// inside the FMAs, no barrier). With W = L11^-1 the panel below the block, the block rows of the
// right-hand sides and their transposed counterparts all become 16x16 matrix-core products, so a
// 16-column block costs a handful of workgroup barriers instead of three per pivot.
// Sblk = &S[j0 * ns + j0] (LDS, row pitch ns), Wblk: 16 x 16, row pitch 17 (LDS).
__device__ __forceinline__ bool chol16_and_inverse(double* Sblk, const int ns, double* Wblk, const int lane) {
  const int r = lane & 15;
  double acc[16], w[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = Sblk[r * ns + c];
  // Left-looking Cholesky, one row per lane of every 16-lane DPP row (the four rows of the wavefront
  // repeat it), fused with the forward substitution of the unit vectors (lane c: column c of
  // W = L11^-1): step j takes row j of L from lane j inside the FMAs (rb_chol_inv, kernels_dpp.hpp)
  const bool bad = rb_chol_inv<16>(r, acc, w);
  {  // every lane stores (lanes >= 16, replicas, into the pad column): a store under a lane
     // predicate makes the compiler sink the W recurrence behind it, away from the broadcasts
    const int wc = lane < 16 ? lane : 16;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) Wblk[rr * 17 + wc] = w[rr];
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c <= r) Sblk[r * ns + c] = acc[c];
  }
  return bad;
}

template <bool STRICT>
__global__ void separator_generic(Dims d, int l, const double* __restrict__ AB, double* F, double* z,
                                  int* __restrict__ info, double* __restrict__ rec, double* scratch = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = d.n, m = d.m, w = d.w, N = d.N;
  const int b = blockIdx.y;
  const int half = 1 << l, base = blockIdx.x * (2 << l), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const int ns = n + 1, ncols = 2 * n + 1;
  const int xs = ncols;
  double* S = scratch ? scratch + ((size_t)b * gridDim.x + blockIdx.x) * ((size_t)n * ns + (size_t)n * xs) : sm;
  double* X = S + (size_t)n * ns;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;

  const double* ab = AB + ((size_t)b * N + s) * n * w;
  const double* Es = Fblk(F, d, b, l, s);
  const double* Es1 = Fblk(F, d, b, l, s + 1);
  const double* Fas = a >= 0 ? Fblk(F, d, b, a, s) : Es;
  const double* Fbs1 = bb >= 0 ? Fblk(F, d, b, bb, s + 1) : Es1;
  const double* zsl = z + ((size_t)b * N + s) * d.rows;
  double* zs1 = z + ((size_t)b * N + s + 1) * d.rows;
  SEG_INIT();

  // ---- P1: inner products. Row i of A_s, B_s against the state / input rows of knot s,
  //      minus the state rows of knot s+1 (its coupling block is [-I; 0]).
  for (int i = wave; i < n; i += nwave) {
    const double* arow = ab + i * w;
    for (int j = lane; j < n; j += 64) {
      double acc = 0.0;
      for (int k = 0; k < n; ++k) acc = mad<STRICT>(arow[k], Es[(n + k) * n + j], acc);
      for (int k = 0; k < m; ++k) acc = mad<STRICT>(arow[n + k], Es[(2 * n + k) * n + j], acc);
      S[i * ns + j] = acc - Es1[(n + i) * n + j];
      double acc2 = 0.0;
      for (int k = 0; k < n; ++k) acc2 = mad<STRICT>(arow[k], Fas[(n + k) * n + j], acc2);
      for (int k = 0; k < m; ++k) acc2 = mad<STRICT>(arow[n + k], Fas[(2 * n + k) * n + j], acc2);
      X[i * xs + j] = acc2;
      X[i * xs + n + j] = -Fbs1[(n + i) * n + j];
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double* arow = ab + i * w;
    double acc = -zs1[i];  // beta = -1 on the old lambda entry (nested_dissection.c:125)
    // same order of operations as the scalar loop (k ascending over [A | B] against z.x | z.u, which
    // are contiguous), operands fetched sixteen pairs at a time
    const double zlast = zs1[n + i];
    for (int k0 = 0; k0 < w; k0 += 16) {
      double av[16], zv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = k0 + u < w ? k0 + u : w - 1;
        av[u] = arow[k];
        zv[u] = zsl[n + k];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (k0 + u < w) acc = mad<STRICT>(av[u], zv[u], acc);
    }
    X[i * xs + 2 * n] = acc - zlast;
  }
  SEG(40);
  __syncthreads();
  SEG(41);

  // ---- P2: lower Cholesky in LDS, right-looking: finish column j, then subtract its outer
  //      product from the remaining lower triangle (wavefront per column c, lane per row i >= c).
  for (int j = 0; j < n; ++j) {
    const double pivot = S[j * ns + j];
    if (!(pivot > 0.0)) {  // uniform: every thread reads the same LDS word
      if (threadIdx.x == 0) flag_failure(info, d, b);
      break;
    }
    const double root = sqrt(pivot);
    __syncthreads();
    for (int i = j + threadIdx.x; i < n; i += blockDim.x) S[i * ns + j] /= root;
    __syncthreads();
    for (int c = j + 1 + wave; c < n; c += nwave) {
      const double lcj = S[c * ns + j];
      for (int i = c + lane; i < n; i += 64) S[i * ns + c] = mad<STRICT>(-S[i * ns + j], lcj, S[i * ns + c]);
    }
    __syncthreads();
  }

  // ---- P3: L Y = X (forward), then L' X = Y (transposed), all 2n+1 columns at once.
  // Per pivot j the scaled row j is kept in registers (one value per lane and 64-column chunk),
  // and each wavefront updates its rows two at a time so that the LDS round trips overlap.
  constexpr int CH = 5;  // column chunks of 64 a lane may own: up to 2n+1 = 320 columns
  auto sweep_rows = [&](int j, int i_begin, int i_end, bool transposed) {
    double xj[CH];
#pragma unroll
    for (int q = 0; q < CH; ++q) { const int c = lane + 64 * q; xj[q] = c < ncols ? X[j * xs + c] : 0.0; }
    int i = i_begin + wave;
    for (; i + nwave < i_end; i += 2 * nwave) {
      const int i2 = i + nwave;
      const double l1 = transposed ? S[j * ns + i] : S[i * ns + j];
      const double l2 = transposed ? S[j * ns + i2] : S[i2 * ns + j];
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int c = lane + 64 * q;
        if (c < ncols) {
          const double a1 = X[i * xs + c], a2 = X[i2 * xs + c];
          X[i * xs + c] = mad<STRICT>(-l1, xj[q], a1);
          X[i2 * xs + c] = mad<STRICT>(-l2, xj[q], a2);
        }
      }
    }
    for (; i < i_end; i += nwave) {
      const double l1 = transposed ? S[j * ns + i] : S[i * ns + j];
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        const int c = lane + 64 * q;
        if (c < ncols) X[i * xs + c] = mad<STRICT>(-l1, xj[q], X[i * xs + c]);
      }
    }
  };
  for (int j = 0; j < n; ++j) {
    const double piv = S[j * ns + j];
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) X[j * xs + c] = X[j * xs + c] / piv;
    __syncthreads();
    const int iend = n;
    if (ncols <= 64 * CH) {
      sweep_rows(j, j + 1, iend, false);
    } else {
      for (int i = j + 1 + wave; i < iend; i += nwave) {
        const double lij = S[i * ns + j];
        for (int c = lane; c < ncols; c += 64) X[i * xs + c] = mad<STRICT>(-lij, X[j * xs + c], X[i * xs + c]);
      }
    }
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; --j) {
    const double piv = S[j * ns + j];
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) X[j * xs + c] = X[j * xs + c] / piv;
    __syncthreads();
    const int ibeg = 0;
    if (ncols <= 64 * CH) {
      sweep_rows(j, ibeg, j, true);
    } else {
      for (int i = ibeg + wave; i < j; i += nwave) {
        const double lji = S[j * ns + i];
        for (int c = lane; c < ncols; c += 64) X[i * xs + c] = mad<STRICT>(-lji, X[j * xs + c], X[i * xs + c]);
      }
    }
    __syncthreads();
  }


  // ---- store into the lambda rows of knot s+1 (columns l, a, bb) and of the rhs
  // With records (fast mode without KEEP: boundary-first schedule, solution by back-substitution)
  // the lambda rows of the factor array are dead data: nothing that reaches a state / input row or a
  // record ever reads them (the boundary Schur pass takes f_a, f_bb from the record), so the
  // Cholesky factor and the two blocks are not written there -- 96 of 160 KB per separator at
  // (64, 16), and the workgroup's LDS is only released when its stores have drained.
  if (!rec) {
    double* outS = Fblk(F, d, b, l, s + 1);
    double* outa = a >= 0 ? Fblk(F, d, b, a, s + 1) : nullptr;
    double* outb = bb >= 0 ? Fblk(F, d, b, bb, s + 1) : nullptr;
    for (int i = wave; i < n; i += nwave)
      for (int c = lane; c < n; c += 64) {
        outS[i * n + c] = S[i * ns + c];
        if (outa) outa[i * n + c] = X[i * xs + c];
        if (outb) outb[i * n + c] = X[i * xs + n + c];
      }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) zs1[i] = X[i * xs + 2 * n];
  if (rec) {
    // record f_a | f_bb | z_sep for the back-substitution (backsub_*_generic): the lambda rows of
    // knot s+1 written above are updated again by later levels when that knot is a boundary knot
    double* myrec = rec + ((size_t)b * d.N + s) * (2 * n * n + n);
    for (int i = wave; i < n; i += nwave)
      for (int c = lane; c < n; c += 64) {
        if (a >= 0) myrec[i * n + c] = X[i * xs + c];
        if (bb >= 0) myrec[n * n + i * n + c] = X[i * xs + n + c];
      }
    for (int i = threadIdx.x; i < n; i += blockDim.x) myrec[2 * n * n + i] = X[i * xs + 2 * n];
  }
#ifdef NDLQR_SEGTIME
  __builtin_amdgcn_s_waitcnt(0);
#endif
  SEG(44);
}

// ------------------------------------------------------------------------------------- Schur update
// grid (ceil(N*rows*n / 256), batch), block 256. One thread per element (knot i, row r, col c):
//   g(i, p) -= E(i) * f_p   for the live outer columns p in {a, bb}, and the rhs (c == 0).
// Left-half knots own column a already (read-modify-write) and get column bb created
// (written, not accumulated); right-half knots the other way round.
// boundary != 0: only the first and the last knot of every level-l subtree (what the next
// separators read; the other knots get their solution from the back-substitution): the grid
// then covers 2 * (N >> (l+1)) knots.
template <bool STRICT>
__global__ void schur_generic(Dims d, int l, double* F, double* z, int boundary, const double* recs = nullptr) {
  const int n = d.n, rows = d.rows, N = d.N;
  const int b = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nknots = boundary ? 2 * (N >> (l + 1)) : N;
  if (e >= (long)nknots * rows * n) return;
  const int ik = (int)(e / (rows * n));
  const int i = boundary ? (ik >> 1) * (2 << l) + ((ik & 1) ? (2 << l) - 1 : 0) : ik;
  const int rem = (int)(e - (long)ik * rows * n);
  const int r = rem / n, c = rem - r * n;
  const int half = 1 << l;
  const int base = (i >> (l + 1)) << (l + 1), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const bool left = i <= s;
  const bool calc_lambda = (i == 0) || (i & (half - 1)) != 0;  // nested_dissection.c:173-177
  // recs (boundary pass of the record-based schedule): f_a, f_bb come from the separator's record
  // and the lambda rows, dead data there, are left alone
  if (r < n && recs) return;
  if (r < n && !calc_lambda) {
    // lambda rows are not updated here; a freshly created block gets explicit zeros there,
    // except at knot s+1 whose lambda rows already hold f_a / f_bb from the separator kernel.
    if (i != s + 1) {
      if (a >= 0 && !left) Fblk(F, d, b, a, i)[r * n + c] = 0.0;
      if (bb >= 0 && left) Fblk(F, d, b, bb, i)[r * n + c] = 0.0;
    }
    return;
  }
  const double* myrec = recs ? recs + ((size_t)b * N + s) * (2 * (size_t)n * n + n) : nullptr;
  const double* Erow = Fblk(F, d, b, l, i) + r * n;
  if (a >= 0) {
    const double* f = myrec ? myrec : Fblk(F, d, b, a, s + 1);  // lambda rows = f_a (n x n, row-major)
    double* g = Fblk(F, d, b, a, i) + r * n + c;
    double acc = left ? *g : 0.0;
    for (int k = 0; k < n; ++k) acc = mad<STRICT>(-Erow[k], f[k * n + c], acc);
    *g = acc;
  }
  if (bb >= 0) {
    const double* f = myrec ? myrec + (size_t)n * n : Fblk(F, d, b, bb, s + 1);
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

// ------------------------------------------------------------------------------------- device-side packing
// The data movement of ndlqr_InitializeWithLQRProblem (src/solver.c:122-194) on the device, for
// producers that already hold the problem in HBM in the flat reference layout (A [N][n*n] and
// B [N][n*m] column-major, Q,q,d [N][n], R,r [N][m], x0 [n] per problem): transposes A, B into the
// row-major [A | B] input, copies the diagonals and builds the negated right-hand side.
// grid (N, batch), any block size.
// du: the caller's dimensions (layout of the flat arrays), d: the device layout -- the same, or a larger block size
// that the problem is zero-padded into (ndlqr_hip.hip, "padded shapes": the pad entries are set once and never touched).
// KP consecutive knots per workgroup, every thread's KP loads of a round in flight before its first store: with one knot per
// workgroup and a load or two per thread the launch ran at (bytes in flight) / (memory latency) = 3 TB/s -- 330 us per
// 1024 x (12,4,256), a third of an iteration of a loop that replaces the whole problem. grid (N / KP, batch).
template <int KP>
static __global__ void pack_flat_generic(Dims du, Dims d, const double* __restrict__ A, const double* __restrict__ B,
                                  const double* __restrict__ Q, const double* __restrict__ R,
                                  const double* __restrict__ q, const double* __restrict__ r,
                                  const double* __restrict__ dd, const double* __restrict__ x0,
                                  double* __restrict__ AB, double* __restrict__ QR, double* __restrict__ rhs) {
  const int k0 = blockIdx.x * KP, b = blockIdx.y;
  const int n = du.n, m = du.m, N = du.N;
  const size_t pk0 = (size_t)b * N + k0;
  for (int e = threadIdx.x; e < n * du.w; e += blockDim.x) {
    const int i = e / du.w, j = e - i * du.w;
    const double* src = j < n ? A + pk0 * n * n + (i + n * j) : B + pk0 * n * m + (i + n * (j - n));
    const size_t sstr = j < n ? (size_t)n * n : (size_t)n * m;
    double* dst = AB + pk0 * d.n * d.w + (i * d.w + (j < n ? j : d.n + (j - n)));
    double v[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) v[kk] = src[kk * sstr];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) dst[(size_t)kk * d.n * d.w] = v[kk];
  }
  for (int e = threadIdx.x; e < du.w; e += blockDim.x) {
    const double* src = e < n ? Q + pk0 * n + e : R + pk0 * m + (e - n);
    const int sstr = e < n ? n : m;
    double* dst = QR + pk0 * d.w + (e < n ? e : d.n + (e - n));
    double v[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) v[kk] = src[kk * sstr];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) dst[kk * d.w] = v[kk];
  }
  for (int e = threadIdx.x; e < du.rows; e += blockDim.x) {
    double* dst = rhs + pk0 * d.rows + (e < n ? e : (e < 2 * n ? d.n + (e - n) : 2 * d.n + (e - 2 * n)));
    double v[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const int k = k0 + kk;
      const size_t pk = pk0 + kk;
      if (e < n) v[kk] = k == 0 ? -x0[(size_t)b * n + e] : -dd[(pk - 1) * n + e];
      else if (e < 2 * n) v[kk] = -q[pk * n + (e - n)];
      else v[kk] = k < N - 1 ? -r[pk * m + (e - 2 * n)] : 0.0;
    }
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) dst[kk * d.rows] = v[kk];
  }
}

// The same for larger blocks, one knot per workgroup of 256 threads: the column-major A_k | B_k come in as they lie (whole
// lines) and leave row-major through LDS -- column j of [A | B] at pack_tile[P j ..], pitch P = n | 1 (odd: the transposed
// reads of a wavefront, stride P doubles, fall into distinct banks). The direct form reads with a stride of n doubles
// between lanes: at (64,16) every lane a line of its own, 2.3 TB/s. grid (N, batch), dynamic LDS P (n + m) doubles.
static __global__ __launch_bounds__(256) void pack_flat_tiled(Dims du, Dims d, const double* __restrict__ A,
                                                              const double* __restrict__ B, const double* __restrict__ Q,
                                                              const double* __restrict__ R, const double* __restrict__ q,
                                                              const double* __restrict__ r, const double* __restrict__ dd,
                                                              const double* __restrict__ x0, double* __restrict__ AB,
                                                              double* __restrict__ QR, double* __restrict__ rhs) {
  extern __shared__ __attribute__((aligned(16))) double pack_tile[];
  const int k = blockIdx.x, b = blockIdx.y;
  const int n = du.n, m = du.m, N = du.N, P = n | 1;
  const size_t pk = (size_t)b * N + k;
  const double* Ak = A + pk * n * n;
  const double* Bk = B + pk * n * m;
  for (int e = threadIdx.x; e < n * du.w; e += blockDim.x) {  // (B_k follows A_k's columns: column j of [A | B] = entries n j ..)
    const int j = e / n, i = e - j * n;
    pack_tile[i + P * j] = j < n ? Ak[e] : Bk[e - n * n];
  }
  // the vectors meanwhile (independent of the tile)
  double* qr = QR + pk * d.w;
  for (int e = threadIdx.x; e < du.w; e += blockDim.x)
    qr[e < n ? e : d.n + (e - n)] = e < n ? Q[pk * n + e] : R[pk * m + (e - n)];
  double* z = rhs + pk * d.rows;
  for (int e = threadIdx.x; e < du.rows; e += blockDim.x) {
    double v;
    if (e < n) v = k == 0 ? -x0[(size_t)b * n + e] : -dd[(pk - 1) * n + e];
    else if (e < 2 * n) v = -q[pk * n + (e - n)];
    else v = k < N - 1 ? -r[pk * m + (e - 2 * n)] : 0.0;
    z[e < n ? e : (e < 2 * n ? d.n + (e - n) : 2 * d.n + (e - 2 * n))] = v;
  }
  __syncthreads();
  double* ab = AB + pk * d.n * d.w;
  for (int e = threadIdx.x; e < n * du.w; e += blockDim.x) {
    const int i = e / du.w, j = e - i * du.w;
    ab[i * d.w + (j < n ? j : d.n + (j - n))] = pack_tile[i + P * j];
  }
}

// Padded shapes: everything of the device inputs that is not a real entry, once, at context creation: zero
// couplings, unit weights (dummy states / inputs with Q = R = 1), zero right-hand side -- the dummies solve to exactly
// zero, S-bar gains a unit diagonal block, nothing else changes. grid (N, batch).
static __global__ void pad_fill_generic(Dims d, double* __restrict__ AB, double* __restrict__ QR, double* __restrict__ rhs) {
  const size_t pk = (size_t)blockIdx.y * d.N + blockIdx.x;
  for (int e = threadIdx.x; e < d.n * d.w; e += blockDim.x) AB[pk * d.n * d.w + e] = 0.0;
  for (int e = threadIdx.x; e < d.w; e += blockDim.x) QR[pk * d.w + e] = 1.0;
  for (int e = threadIdx.x; e < d.rows; e += blockDim.x) rhs[pk * d.rows + e] = 0.0;
}

// Packed inputs in the CALLER's block size (the host layout of ndlqr_hip_upload_inputs, staged in HBM) into the padded
// device arrays of problems [p0, p0 + count). Null sAB: the right-hand side alone. grid (N, count).
static __global__ void pad_inputs_generic(Dims du, Dims d, const int p0, const double* __restrict__ sAB,
                                          const double* __restrict__ sQR, const double* __restrict__ srhs,
                                          double* __restrict__ AB, double* __restrict__ QR, double* __restrict__ rhs) {
  const size_t sk = (size_t)blockIdx.y * du.N + blockIdx.x, pk = ((size_t)p0 + blockIdx.y) * d.N + blockIdx.x;
  const int n = du.n;
  if (sAB) {
    for (int e = threadIdx.x; e < n * du.w; e += blockDim.x) {
      const int i = e / du.w, j = e - i * du.w;
      AB[pk * d.n * d.w + i * d.w + (j < n ? j : d.n + (j - n))] = sAB[sk * n * du.w + e];
    }
    for (int e = threadIdx.x; e < du.w; e += blockDim.x) QR[pk * d.w + (e < n ? e : d.n + (e - n))] = sQR[sk * du.w + e];
  }
  for (int e = threadIdx.x; e < du.rows; e += blockDim.x)
    rhs[pk * d.rows + (e < n ? e : (e < 2 * n ? d.n + (e - n) : 2 * d.n + (e - 2 * n)))] = srhs[sk * du.rows + e];
}

// One problem's solution blocks [N][2 n + m] (caller's block size, the u slot of the last knot included) out of
// the padded device array. grid (N).
static __global__ void unpad_blocks_generic(Dims du, Dims d, const double* __restrict__ z, double* __restrict__ dst) {
  const int k = blockIdx.x, n = du.n;
  for (int e = threadIdx.x; e < du.rows; e += blockDim.x)
    dst[(size_t)k * du.rows + e] = z[(size_t)k * d.rows + (e < n ? e : (e < 2 * n ? d.n + (e - n) : 2 * d.n + (e - 2 * n)))];
}

// The same as a STREAMING kernel: few workgroups (the launch is resident at once, so that kernels of the other buffer
// set's stream run beside it), grid-stride, four independent loads in flight per thread -- made to read q, r, d, x0
// straight from PINNED HOST memory over the host link (no staging copy; a copy engine transfer in both directions on
// one stream serialises the two-deep step pipeline on this platform, tools/ubench/copy_overlap.hip). Any of q / r / d
// may be null: that part of the right-hand side stays as it is (an MPC iteration often replaces x0 alone).
__device__ __forceinline__ void pack_rhs_one(const Dims& du, const Dims& d, const int which, const size_t g,
                                             const double v, double* __restrict__ rhs) {
  const size_t n = du.n, m = du.m, rows = d.rows, N = du.N, np = d.n;  // rows, np: the (possibly padded) device layout
  if (which == 0) {         // q
    const size_t pk = g / n, e = g - pk * n;
    rhs[pk * rows + np + e] = -v;
  } else if (which == 1) {  // r (the last knot's input slot stays 0)
    const size_t pk = g / m, e = g - pk * m;
    rhs[pk * rows + 2 * np + e] = (pk % N) < N - 1 ? -v : 0.0;
  } else if (which == 2) {  // d_k is the lambda block of knot k + 1
    const size_t pk = g / n, e = g - pk * n;
    if ((pk % N) < N - 1) rhs[(pk + 1) * rows + e] = -v;
  } else {                  // x0: the lambda block of knot 0
    const size_t b = g / n, e = g - b * n;
    rhs[b * N * rows + e] = -v;
  }
}

static __global__ __launch_bounds__(256) void pack_rhs_stream_generic(Dims du, Dims d, const double* __restrict__ q,
                                                                      const double* __restrict__ r,
                                                                      const double* __restrict__ dd,
                                                                      const double* __restrict__ x0,
                                                                      double* __restrict__ rhs) {
  const size_t nq = (size_t)du.batch * du.N * du.n, nr = (size_t)du.batch * du.N * du.m, nx = (size_t)du.batch * du.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double* src[4] = {q, r, dd, x0};
  const size_t cnt[4] = {nq, nr, nq, nx};
#pragma unroll
  for (int which = 0; which < 4; ++which) {
    const double* p = src[which];
    if (!p) continue;
    const size_t n = cnt[which];
    size_t i = t0;
    for (; i + 3 * stride < n; i += 4 * stride) {
      const double a = p[i], b = p[i + stride], c = p[i + 2 * stride], e = p[i + 3 * stride];
      pack_rhs_one(du, d, which, i, a, rhs);
      pack_rhs_one(du, d, which, i + stride, b, rhs);
      pack_rhs_one(du, d, which, i + 2 * stride, c, rhs);
      pack_rhs_one(du, d, which, i + 3 * stride, e, rhs);
    }
    for (; i < n; i += stride) pack_rhs_one(du, d, which, i, p[i], rhs);
  }
}

// Solutions [batch][N][2n+m] (the unused trailing u_N slot included) -> [batch][nvars] packed, the
// layout of ndlqr_CopyBatchSolutions, in device memory. grid (N, batch).
// grid (pack_solutions_chunks(du), batch), block 256: a workgroup takes PACK_SOLN_CHUNK consecutive entries of a problem's
// solution vector, four per thread, all four loads in flight before the first store. (One 64-thread workgroup per knot --
// 28 loads in flight -- ran at 1 TB/s: 59 us for the 59 MB of 1024 x (12,4,256).)
constexpr int PACK_SOLN_CHUNK = 1024;
static inline unsigned pack_solutions_chunks(const Dims& du) {
  return (unsigned)(((size_t)du.rows * du.N - du.m + PACK_SOLN_CHUNK - 1) / PACK_SOLN_CHUNK);
}
static __global__ __launch_bounds__(256) void pack_solutions_generic(Dims du, Dims d, const double* __restrict__ z,
                                                                     double* __restrict__ dst) {
  const int b = blockIdx.y, n = du.n, rows = du.rows;
  const int nvars = rows * du.N - du.m;
  const double* zb = z + (size_t)b * d.N * d.rows;
  double v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int o = blockIdx.x * PACK_SOLN_CHUNK + threadIdx.x + 256 * u, oc = o < nvars ? o : nvars - 1;
    const int k = oc / rows, e = oc - k * rows;
    v[u] = zb[(size_t)k * d.rows + (e < n ? e : (e < 2 * n ? d.n + (e - n) : 2 * d.n + (e - 2 * n)))];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int o = blockIdx.x * PACK_SOLN_CHUNK + threadIdx.x + 256 * u;
    if (o < nvars) dst[(size_t)b * nvars + o] = v[u];
  }
}

// A slice of the solutions, packed for the host: knots [knot0, knot0 + nknots) of every problem, of each knot the
// blocks selected by `blocks` (bit 0: lambda, 1: state, 2: input), in the reference's order -> [batch][nknots][width],
// width = n * (bits 0, 1) + m * (bit 2). What an MPC loop consumes of a solve is u of knot 0 (32 KB per 1024
// problems of (12,4) against 59 MB for every lambda, x, u). The input slot of the last knot is not part of the
// solution (src/solver.c:64): it comes out as stored (zero). grid (nknots, batch).
static __global__ void pack_selection_generic(Dims du, Dims d, const int knot0, const int nknots, const unsigned blocks,
                                              const double* __restrict__ z, double* __restrict__ dst) {
  const int kk = blockIdx.x, b = blockIdx.y, n = du.n, m = du.m;
  const int nl = (blocks & 1u) ? n : 0, nxs = (blocks & 2u) ? n : 0, nu = (blocks & 4u) ? m : 0;
  const int width = nl + nxs + nu;
  const double* zk = z + ((size_t)b * d.N + knot0 + kk) * d.rows;
  double* out = dst + ((size_t)b * nknots + kk) * width;
  for (int e = threadIdx.x; e < width; e += blockDim.x) {
    int src;
    if (e < nl) src = e;
    else if (e < nl + nxs) src = d.n + (e - nl);
    else src = 2 * d.n + (e - nl - nxs);
    out[e] = zk[src];
  }
}

// The right-hand side exists once per buffer set of the solve pipeline (ndlqr_hip_step_async replaces it per step):
// copies the parts of `mask` -- bit 0: q (state rows), 1: r (input rows), 2: d (lambda rows of the knots >= 1),
// 3: x0 (lambda rows of knot 0) -- from one set's copy to the other's. grid (N, batch).
static __global__ void copy_rhs_parts_generic(Dims d, const unsigned mask, const double* __restrict__ src,
                                              double* __restrict__ dst) {
  const int k = blockIdx.x;
  const size_t base = ((size_t)blockIdx.y * d.N + k) * d.rows;
  for (int e = threadIdx.x; e < d.rows; e += blockDim.x) {
    const unsigned bit = e < d.n ? (k == 0 ? 8u : 4u) : (e < 2 * d.n ? 1u : 2u);
    if (mask & bit) dst[base + e] = src[base + e];
  }
}

// ------------------------------------------------------------------------------------- rhs-only sweep
// Factor / solve split (the reference cannot separate them, docs/rslqr_usage.dox):
// with the complete factor array kept on the device (NDLQR_FLAG_KEEP_FACT) a new right-hand side
// only needs the reference's solution sweep (src/solve.c:137-182) -- per level: inner product
// with the rhs, solve with the cached Cholesky factor of S-bar, propagate with column l of the
// factorisation -- preceded by the rhs part of the leaf phase. Same operations and order as the
// rhs column of the fused solve, so the results are identical to a full re-solve.
template <bool STRICT>
__global__ void rhs_leaf_generic(Dims d, const double* __restrict__ QR, const double* __restrict__ rhs,
                                 double* __restrict__ z) {
  const int k = blockIdx.x, b = blockIdx.y;
  const int n = d.n, w = d.w, rows = d.rows, N = d.N;
  const double* qr = QR + ((size_t)b * N + k) * w;
  const double* r0 = rhs + ((size_t)b * N + k) * rows;
  double* zk = z + ((size_t)b * N + k) * rows;
  const bool last = (k == N - 1);
  for (int i = threadIdx.x; i < rows; i += blockDim.x) {
    double v = r0[i];
    if (k == 0 && i < n) {
      v = mad<STRICT>(-qr[i], r0[i], -r0[n + i]);
    } else if (k == 0 && i < 2 * n) {
      v = -r0[i - n];
    } else if (i >= n && (i < 2 * n || !last)) {
      const double sc = qr[i - n] / sqrt(qr[i - n]);
      v = (v / sc) / sc;
    }
    zk[i] = v;
  }
}

// grid (N >> (l+1), batch), block 64 (one wavefront), dynamic LDS n (n+1) + n doubles: the cached
// factor is staged in LDS first (whole rows, coalesced) so that the substitutions do not pay a
// global-memory round trip per pivot. staged == 0 (blocks whose factor does not fit the LDS: beyond ~140 states):
// dynamic LDS n doubles, the factor is read where it lies.
template <bool STRICT>
__global__ void rhs_separator_generic(Dims d, int l, const double* __restrict__ AB,
                                      const double* __restrict__ F, double* z, const int staged = 1) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = d.n, m = d.m, w = d.w, N = d.N, b = blockIdx.y;
  const int half = 1 << l, base = blockIdx.x * (2 << l), s = base + half - 1;
  const double* ab = AB + ((size_t)b * N + s) * n * w;
  const double* zsl = z + ((size_t)b * N + s) * d.rows;
  double* zs1 = z + ((size_t)b * N + s + 1) * d.rows;
  const double* L = Fblk(F, d, b, l, s + 1);  // lambda rows: Cholesky factor of S-bar, row-major
  const int ns = staged ? n + 1 : n;
  const double* Ls = staged ? sm : L;
  double* v = staged ? sm + n * ns : sm;
  if (staged) {
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
      const int i = e / n, j = e - i * n;
      sm[i * ns + j] = L[e];
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double* arow = ab + i * w;
    double acc = -zs1[i];
    for (int k = 0; k < n; ++k) acc = mad<STRICT>(arow[k], zsl[n + k], acc);
    for (int k = 0; k < m; ++k) acc = mad<STRICT>(arow[n + k], zsl[2 * n + k], acc);
    v[i] = acc - zs1[n + i];
  }
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    if (threadIdx.x == 0) v[j] = v[j] / Ls[j * ns + j];
    __syncthreads();
    const double vj = v[j];
    for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) v[i] = mad<STRICT>(-Ls[i * ns + j], vj, v[i]);
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; --j) {
    if (threadIdx.x == 0) v[j] = v[j] / Ls[j * ns + j];
    __syncthreads();
    const double vj = v[j];
    for (int i = threadIdx.x; i < j; i += blockDim.x) v[i] = mad<STRICT>(-Ls[j * ns + i], vj, v[i]);
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) zs1[i] = v[i];
}

// grid (ceil(N*rows / 256), batch), block 256: z(i)[r] -= F(i, l)(r, :) . z_sep ; one thread per
// row, whole-row 16-byte loads when the row length allows it.
template <bool STRICT>
__global__ void rhs_update_generic(Dims d, int l, const double* __restrict__ F, double* z) {
  const int n = d.n, rows = d.rows, N = d.N, b = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * rows) return;
  const int i = e / rows, r = e - i * rows;
  const int half = 1 << l;
  const int s = ((i >> (l + 1)) << (l + 1)) + half - 1;
  const bool calc_lambda = (i == 0) || (i & (half - 1)) != 0;
  if (r < n && !calc_lambda) return;
  const double* Erow = Fblk(F, d, b, l, i) + (size_t)r * n;
  const double* zsep = z + ((size_t)b * N + s + 1) * rows;
  double* g = z + ((size_t)b * N + i) * rows + r;
  double acc = *g;
  if ((n & 1) == 0) {
    const double2* E2 = reinterpret_cast<const double2*>(Erow);
    for (int k = 0; k < n / 2; ++k) {
      const double2 ev = E2[k];
      acc = mad<STRICT>(-ev.x, zsep[2 * k], acc);
      acc = mad<STRICT>(-ev.y, zsep[2 * k + 1], acc);
    }
  } else {
    for (int k = 0; k < n; ++k) acc = mad<STRICT>(-Erow[k], zsep[k], acc);
  }
  *g = acc;
}

// ------------------------------------------------------------------------------------- back-substitution
// Runtime-sized form of backsub_small (kernels_small.hpp): multipliers level by level, top-down,
//   y_s = z_sep(s) - f_a(s) y_A - f_bb(s) y_B     (record of s; y_A, y_B final in z(A+1), z(B+1))
// written to the lambda rows of knot s+1, then states and inputs of every knot from the problem
// data. Fast mode without KEEP: the Schur passes then only touch the boundary knots.
//   multipliers: grid (N >> (l+1), batch), block 64, once per level K-1 .. 0
//   states:      grid (ceil(N * rows / 256), batch), block 256
static __global__ void backsub_multipliers_generic(Dims d, int l, const double* __restrict__ recs, double* z) {
  // one wavefront per separator: lanes walk the columns of a record row (coalesced), the row sum
  // is a wavefront reduction
  const int n = d.n, rows = d.rows, N = d.N, b = blockIdx.y, lane = threadIdx.x;
  const int T = 2 << l, base = blockIdx.x * T, s = base + (1 << l) - 1;
  const double* rc = recs + ((size_t)b * N + s) * (2 * n * n + n);
  const bool hasA = base > 0, hasB = base + T < N;
  const double* yA = z + ((size_t)b * N + base) * rows;      // y_A lives in the lambda rows of knot A+1 = base
  const double* yB = z + ((size_t)b * N + base + T) * rows;
  double* out = z + ((size_t)b * N + s + 1) * rows;
  // eight record rows per load round (a row at a time is a memory round trip per row); per row the
  // same order of summation as before
  for (int r0 = 0; r0 < n; r0 += 8) {
    double part[8], zr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      part[u] = 0.0;
      zr[u] = rc[2 * n * n + (r0 + u < n ? r0 + u : n - 1)];
    }
    for (int c = lane; c < n; c += 64) {
      double fa[8], fb[8], ya = 0.0, yb = 0.0;
      if (hasA) {  // uniform
        ya = yA[c];
#pragma unroll
        for (int u = 0; u < 8; ++u) fa[u] = rc[(size_t)(r0 + u < n ? r0 + u : n - 1) * n + c];
      }
      if (hasB) {
        yb = yB[c];
#pragma unroll
        for (int u = 0; u < 8; ++u) fb[u] = rc[(size_t)n * n + (size_t)(r0 + u < n ? r0 + u : n - 1) * n + c];
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
      if (lane == 0 && r0 + u < n) out[r0 + u] = zr[u] - p;
    }
  }
}

static __global__ void backsub_states_generic(Dims d, const double* __restrict__ AB, const double* __restrict__ QR,
                                              const double* __restrict__ rhs, double* z) {
  const int n = d.n, rows = d.rows, w = d.w, N = d.N, b = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * rows) return;
  const int k = e / rows, r = e - k * rows;
  if (r < n && k > 0) return;  // lambda rows of knots >= 1 are the multipliers already
  const double* r0 = rhs + ((size_t)b * N + k) * rows;
  const double* qr = QR + ((size_t)b * N + k) * w;
  const double* ab = AB + ((size_t)b * N + k) * n * w;
  double* zk = z + ((size_t)b * N + k) * rows;
  const double* yk = zk + rows;  // y_k: lambda rows of knot k+1
  const int col = r < n ? r : r - n;  // column of [A_k | B_k] this row meets
  double dot = 0.0;
  if (k < N - 1 && !(k == 0 && r >= n && r < 2 * n)) {
    for (int c0 = 0; c0 < n; c0 += 16) {  // sixteen operand pairs per load round, same order of summation
      double av[16], yv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int c = c0 + u < n ? c0 + u : n - 1;
        av[u] = ab[(size_t)c * w + col];
        yv[u] = yk[c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (c0 + u < n) dot = fma(av[u], yv[u], dot);
    }
  }
  double out;
  if (r < n) out = fma(-qr[r], r0[r], -r0[n + r]) + dot;                               // knot 0: Q x0 + q + A_0' y_0
  else if (r < 2 * n) out = (k == 0) ? -r0[r - n] : (r0[r] - dot + zk[r - n]) / qr[r - n];  // x_k (zk[.] = y_{k-1})
  else out = (k == N - 1) ? r0[r] : (r0[r] - dot) / qr[r - n];                          // u_k
  zk[r] = out;
}

// ------------------------------------------------------------------------------------- KKT residual
// Secondary witness: the residual of the solution in z against the RAW problem
// (the packed inputs, not the factorisation), per problem:
//   x_0 - x_init;  Q x_k + q_k - lambda_k + A_k' lambda_{k+1};  R u_k + r_k + B_k' lambda_{k+1};
//   A_k x_k + B_k u_k + d_k - x_{k+1}                        (oracle_kkt_residual, oracle/ndlqr_oracle.c)
// z(k) = [lambda_k | x_k | u_k] with lambda_k the multiplier of the dynamics INTO knot k; rhs holds
// -(x_init or d_{k-1}) | -q_k | -r_k. out[b] = ||K z - b||_2, out[batch + b] = ||b||_2.
//   grid (batch), block 256.
static __global__ void kkt_residual_generic(Dims d, const double* __restrict__ AB, const double* __restrict__ QR,
                                     const double* __restrict__ rhs, const double* __restrict__ z,
                                     double* __restrict__ out) {
  const int n = d.n, N = d.N, rows = d.rows, w = d.w, b = blockIdx.x;
  const double* zb = z + (size_t)b * N * rows;
  const double* rb = rhs + (size_t)b * N * rows;
  double res = 0.0, bn = 0.0;
  for (int e = threadIdx.x; e < N * rows; e += blockDim.x) {
    const int k = e / rows, r = e - k * rows;
    const double* zk = zb + (size_t)k * rows;
    const double* ab = AB + ((size_t)b * N + k) * n * w;
    const double* qr = QR + ((size_t)b * N + k) * w;
    double v = 0.0, bv = -rb[(size_t)k * rows + r];
    if (r < n) {
      if (k == 0) {
        v = zk[n + r] - bv;  // x_0 - x_init
      } else {
        const double* zp = zk - rows;                             // knot k-1
        const double* abp = ab - (size_t)n * w + (size_t)r * w;   // row r of [A_{k-1} | B_{k-1}]
        v = bv - zk[n + r];                                       // d_{k-1} - x_k
        for (int j = 0; j < w; ++j) v = fma(abp[j], zp[n + j], v);
      }
    } else if (r < 2 * n) {
      const int i = r - n;
      v = fma(qr[i], zk[n + i], bv) - zk[i];
      if (k < N - 1)
        for (int j = 0; j < n; ++j) v = fma(ab[(size_t)j * w + i], zk[rows + j], v);
    } else if (k < N - 1) {
      const int i = r - 2 * n;
      v = fma(qr[n + i], zk[2 * n + i], bv);
      for (int j = 0; j < n; ++j) v = fma(ab[(size_t)j * w + n + i], zk[rows + j], v);
    } else {
      bv = 0.0;  // the u slot of the last knot is not part of the system
    }
    res = fma(v, v, res);
    bn = fma(bv, bv, bn);
  }
  __shared__ double sres[256], sbn[256];
  sres[threadIdx.x] = res; sbn[threadIdx.x] = bn;
  __syncthreads();
  for (int off = blockDim.x / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { sres[threadIdx.x] += sres[threadIdx.x + off]; sbn[threadIdx.x] += sbn[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[b] = sqrt(sres[0]); out[d.batch + b] = sqrt(sbn[0]); }
}


// ------------------------------------------------------------------------------------- dense helpers
// Device versions of the reference's internal routines, one element / column per thread.
// C = alpha*op(A)*op(B) + beta*C  (linalg_custom.c:20-43): beta first, then k ascending.
static __global__ void dense_gemm(int tA, int tB, int m, int n, int k, double alpha, const double* A, int lda,
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
static __global__ void dense_potrf(int n, double* A, int lda, int* info) {
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

// L L' x = b (linalg_custom.c:113-138); one thread per right-hand side. which = 1: the forward substitution
// alone (L x = b), 2: the transposed one alone (L' x = b) -- clap_LowerTriBackSub.
static __global__ void dense_potrs(int n, int nrhs, const double* L, int ldl, double* B, int ldb, int which) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nrhs) return;
  double* x = B + (size_t)ldb * c;
  if (which != 2)
    for (int j = 0; j < n; ++j) {
      const double xj = x[j] / L[j + (size_t)ldl * j];
      x[j] = xj;
      for (int i = j + 1; i < n; ++i) x[i] = x[i] - L[i + (size_t)ldl * j] * xj;
    }
  if (which != 1)
    for (int j = n - 1; j >= 0; --j) {
      const double xj = x[j] / L[j + (size_t)ldl * j];
      x[j] = xj;
      for (int i = 0; i < j; ++i) x[i] = x[i] - L[j + (size_t)ldl * i] * xj;
    }
}

}  // namespace ndlqr
