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

//   grid (N >> (l+1), batch), block 64 * nwave (nwave >= n / 16, tiles * CT <= 3 nwave, tiles^2 <= MAXS nwave),
//   dynamic LDS = n (n + 1) + n (16 min(CT, 2 n / 16 + 1) + 1) + 17 n + 2 (n + m) + 2 n doubles
//   (the staged [A_s | B_s], n rows of reduced_stage_pitch(w), lies over the first two arrays: host check).
template <int CT>
__global__ void separator_reduced_mfma(Dims d, int l, const double* __restrict__ AB, const double* __restrict__ QR,
                                       const double* __restrict__ rhs, double* red, double* __restrict__ rec,
                                       int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = d.n, w = d.w, N = d.N, rows = d.rows;
  const int b = blockIdx.y;
  const int T = 2 << l, base = blockIdx.x * T, s = base + (1 << l) - 1;
  const bool hasA = base > 0, hasB = base + T < N, leftchild = (base & T) == 0, first = s == 0;
  const int ns = n + 1, tiles = n >> 4, ctl = 2 * tiles + 1;  // column tiles: f_a, f_bb, [z_sep | padding]
  const int ctc = ctl < CT ? ctl : CT, xs = 16 * ctc + 1;
  const int nn = n * n;
  double* S = sm;
  double* X = S + n * ns;
  double* Wd = X + (size_t)n * xs;  // n / 16 blocks of 16 x 17: inverses of the diagonal blocks of L
  double* dq = Wd + n * 17;         // 1 / [Q_s | R_s]  (state entries of knot 0: zero -- its state is fixed)
  double* zc = dq + w;              // rhs(s).xu scaled likewise (state entries of knot 0: -x0)
  double* q1 = zc + w;              // 1 / Q_{s+1}
  double* bz = q1 + n;              // b~
  double* stage = sm;               // [A_s | B_s], pitch P, over S and X until S-bar is formed
  const int P = reduced_stage_pitch(w);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const SepGeom geo = {n, ns, xs, tiles, lane, wave, nwave, li, lk};

  const double* ab = AB + ((size_t)b * N + s) * n * w;   // [A_s | B_s]
  const double* ab1 = ab + (size_t)n * w;                 // [A_{s+1} | B_{s+1}]  (s + 1 <= N - 1)
  const double* qr = QR + ((size_t)b * N + s) * w;
  const double* r0 = rhs + ((size_t)b * N + s) * rows;
  const double* myslot = reduced_slot(red, d, b, s);      // only read for l >= 1
  double* slotA = reduced_slot(red, d, b, hasA ? base - 1 : 1);
  double* slotB = reduced_slot(red, d, b, hasB ? base + T - 1 : 1);
  const int ksteps = w / 4;

  // coupling blocks as they enter the panel and the pushes: r_a(k, j), r_bb(k, j)
  auto ra_at = [&](const int k, const int j) -> double {
    return l == 0 ? -ab[(size_t)k * w + j] * dq[j] : -myslot[2 * nn + k * n + j];
  };
  auto rbb_at = [&](const int k, const int j) -> double {
    return l == 0 ? -q1[k] * ab1[(size_t)j * w + k] : -myslot[3 * nn + k * n + j];
  };

  // ---- stage [A_s | B_s] (16-byte words, coalesced) and the diagonal weights / rhs of knots s, s + 1
  {
    const int words = n * w / 2;
    for (int e0 = 0; e0 < words; e0 += 4 * nthr) {  // four loads in flight per thread
      double2 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * nthr + tid;
        t[u] = reinterpret_cast<const double2*>(ab)[e < words ? e : words - 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * nthr + tid;
        if (e < words) {
          const int row = (2 * e) / w, col = 2 * e - row * w;
          *reinterpret_cast<double2*>(stage + row * P + col) = t[u];
        }
      }
    }
    for (int k = tid; k < w; k += nthr) {
      const bool fx = first && k < n;
      const double inv = 1.0 / qr[k];
      dq[k] = fx ? 0.0 : inv;
      zc[k] = fx ? -r0[k] : r0[n + k] * inv;
    }
    for (int i = tid; i < n; i += nthr) q1[i] = 1.0 / qr[w + i];
  }
  __syncthreads();

  // ---- leafS = [A_s | B_s] diag(dq) [A_s | B_s]' as matrix-core tiles held in the accumulators until
  //      every wavefront has finished reading the staged block (S-bar is written over it)
  constexpr int MAXS = 5;
  mfma_acc_t sacc[MAXS];
#pragma unroll
  for (int idx = 0; idx < MAXS; ++idx) {
    const int item = wave + idx * nwave;
    mfma_acc_t acc = {0.0, 0.0, 0.0, 0.0};
    if (item < tiles * tiles) {
      const int rt = item / tiles, ct = item % tiles;
      const double* arow = stage + (16 * rt + li) * P + lk;
      const double* brow = stage + (16 * ct + li) * P + lk;
      for (int q = 0; q < ksteps; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[4 * q], brow[4 * q] * dq[4 * q + lk], acc, 0, 0, 0);
    }
    sacc[idx] = acc;
  }
  // b~ = [A_s | B_s] zc - z(s+1).lambda - z(s+1).x / Q_{s+1} - gL - gR
  double bt = 0.0;
  if (tid < n) {
    const double* arow = stage + tid * P;
    double acc = -fma(r0[rows + n + tid], q1[tid], r0[rows + tid]);
    for (int k = 0; k < w; ++k) acc = fma(arow[k], zc[k], acc);
    if (l > 0) acc -= myslot[4 * nn + tid] + myslot[4 * nn + n + tid];
    bt = acc;
  }
  __syncthreads();
  if (tid < n) bz[tid] = bt;
#pragma unroll
  for (int idx = 0; idx < MAXS; ++idx) {
    const int item = wave + idx * nwave;
    if (item < tiles * tiles) {
      const int rt = item / tiles, ct = item % tiles;
      double dl[4] = {0.0, 0.0, 0.0, 0.0}, dr[4] = {0.0, 0.0, 0.0, 0.0};
      if (l > 0) {
        const double* p = myslot + (size_t)(16 * rt + lk) * n + 16 * ct + li;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) { dl[gg] = p[4 * gg * n]; dr[gg] = p[nn + 4 * gg * n]; }
      }
      double* dst = S + (16 * rt + lk) * ns + 16 * ct + li;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const int i = 16 * rt + lk + 4 * gg, j = 16 * ct + li;
        dst[4 * gg * ns] = sacc[idx][gg] + (i == j ? q1[i] : 0.0) - dl[gg] - dr[gg];
      }
    }
  }
  __syncthreads();

  sep_cholesky(geo, S, Wd, info, d, b);
  sep_invert(geo, S, Wd);

  double* myrec = rec + ((size_t)b * N + s) * (2 * (size_t)nn + n);

  // ---- the panel, CT column tiles at a time: build, X = W' (W R), record, pushes
  for (int t0 = 0; t0 < ctl; t0 += ctc) {
    const int tc = ctl - t0 < ctc ? ctl - t0 : ctc;
    for (int t = 0; t < tc; ++t) {  // (tile kinds are uniform over the workgroup)
      const int gt = t0 + t;
      double* dst = X + 16 * t;
      if (gt < tiles) {  // r_a, columns 16 gt ..
        for (int e = tid; e < 16 * n; e += nthr) {
          const int i = e >> 4, cl = e & 15;
          dst[i * xs + cl] = hasA ? ra_at(i, 16 * gt + cl) : 0.0;
        }
      } else if (gt < 2 * tiles) {  // r_bb
        const int c0 = 16 * (gt - tiles);
        if (l == 0) {  // rows of A_{s+1} are the columns of r_bb: the row index runs fastest (coalesced)
          for (int e = tid; e < 16 * n; e += nthr) {
            const int cl = e / n, i = e - cl * n;
            dst[i * xs + cl] = hasB ? rbb_at(i, c0 + cl) : 0.0;
          }
        } else {
          for (int e = tid; e < 16 * n; e += nthr) {
            const int i = e >> 4, cl = e & 15;
            dst[i * xs + cl] = hasB ? rbb_at(i, c0 + cl) : 0.0;
          }
        }
      } else {  // [b~ | padding]
        for (int e = tid; e < 16 * n; e += nthr) {
          const int i = e >> 4, cl = e & 15;
          dst[i * xs + cl] = cl == 0 ? bz[i] : 0.0;
        }
      }
    }
    __syncthreads();
    sep_panel_solve(geo, S, Wd, X, tc);

    // record f_a | f_bb | z_sep (what the back-substitution reads)
    for (int i = wave; i < n; i += nwave) {
      for (int cc = lane; cc < 16 * tc; cc += 64) {
        const int gt = t0 + (cc >> 4), cl = cc & 15;
        const double v = X[i * xs + cc];
        if (gt < tiles) { if (hasA) myrec[i * n + 16 * gt + cl] = v; }
        else if (gt < 2 * tiles) { if (hasB) myrec[nn + i * n + 16 * (gt - tiles) + cl] = v; }
        else if (cl == 0) myrec[2 * nn + i] = v;
      }
    }
    // pushes: per column tile of the chunk and output row tile rt, one n-deep product of a coupling
    // block (operand from global memory / L2) with the solved tile (operand from LDS)
    //   f_a tile:  kind 0  DR[A] += r_a' f_a
    //   f_bb tile: kind 0  DL[B] += r_bb' f_bb      kind 1  CA[B] = f_bb' r_a  or  CB[A] = r_a' f_bb
    //   z tile:    kind 0  gR[A] += r_a' z_sep      kind 1  gL[B] += r_bb' z_sep   (column 0 of the tile)
    for (int item = wave; item < tc * 2 * tiles; item += nwave) {
      const int t = item / (2 * tiles), kind = (item / tiles) & 1, rt = item % tiles, gt = t0 + t;
      const int ttype = gt < tiles ? 0 : (gt < 2 * tiles ? 1 : 2);
      const bool use_ra = ttype == 0 || (ttype == 1 && kind == 1) || (ttype == 2 && kind == 0);
      if (ttype == 0 && kind == 1) continue;
      if (use_ra && !hasA) continue;
      if ((!use_ra || (ttype == 1 && kind == 1)) && !hasB) continue;
      const bool xa = ttype == 1 && kind == 1 && leftchild;  // the solved tile is the A operand (CA = f_bb' r_a)
      // destination tile: rows 16 orow + lk + 4 g, columns 16 ocol + li, row pitch n (vectors: column 0 only)
      double* dst;
      int orow, ocol;
      bool vec = false, accumulate = l > 0;
      if (ttype == 0) { dst = slotA + nn; orow = rt; ocol = gt; }
      else if (ttype == 1 && kind == 0) { dst = slotB; orow = rt; ocol = gt - tiles; }
      else if (ttype == 1) {
        accumulate = false;
        if (leftchild) { dst = slotB + 2 * nn; orow = gt - tiles; ocol = rt; }
        else { dst = slotA + 3 * nn; orow = rt; ocol = gt - tiles; }
      } else {
        vec = true; orow = rt; ocol = 0;
        dst = kind == 0 ? slotA + 4 * nn + n : slotB + 4 * nn;
      }
      mfma_acc_t acc = {0.0, 0.0, 0.0, 0.0};
      if (accumulate) {
        if (vec) {
          if (li == 0) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) acc[gg] = dst[16 * orow + lk + 4 * gg];
          }
        } else {
          const double* p = dst + (size_t)(16 * orow + lk) * n + 16 * ocol + li;
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) acc[gg] = p[4 * gg * n];
        }
      }
      const double* xcol = X + lk * xs + 16 * t + li;
      constexpr int CH = 8;
      for (int q0 = 0; q0 < n / 4; q0 += CH) {  // n / 4 k-steps, operand fragments of eight at a time
        double gf[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = q0 + c < n / 4 ? q0 + c : n / 4 - 1;
          gf[c] = use_ra ? ra_at(4 * q + lk, 16 * rt + li) : rbb_at(4 * q + lk, 16 * rt + li);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (q0 + c < n / 4) {  // uniform
            const double xv = xcol[(size_t)4 * (q0 + c) * xs];
            acc = xa ? __builtin_amdgcn_mfma_f64_16x16x4f64(xv, gf[c], acc, 0, 0, 0)
                     : __builtin_amdgcn_mfma_f64_16x16x4f64(gf[c], xv, acc, 0, 0, 0);
          }
        }
      }
      if (vec) {
        if (li == 0) {
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) dst[16 * orow + lk + 4 * gg] = acc[gg];
        }
      } else {
        double* p = dst + (size_t)(16 * orow + lk) * n + 16 * ocol + li;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) p[4 * gg * n] = acc[gg];
      }
    }
    __syncthreads();  // the next chunk overwrites the panel
  }
}

}  // namespace ndlqr
