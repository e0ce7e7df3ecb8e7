// kernels_mfma.hpp -- large-block separator and Schur update on the fp64 matrix cores (gfx950).
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
#include "kernels_generic.hpp"  // chol16_and_inverse

namespace ndlqr {

typedef double mfma_acc_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------- shared phases
// Geometry of a separator workgroup: S-bar / L / W in LDS with row pitch ns = n + 1, ONE chunk of the
// right-hand-side panel with row pitch xs, the inverses of the 16x16 diagonal blocks of L (pitch 17).
struct SepGeom {
  int n, ns, xs, tiles;
  int lane, wave, nwave, li, lk;
};
constexpr int kSepMaxPanelTiles = 3;  // panel tiles a wavefront may own: tiles * CT <= 3 nwave (checked on the host)

// Four k-steps of a tile product: acc += A B with the operand fragments already in registers.
__device__ __forceinline__ mfma_acc_t mfma4(const double (&a)[4], const double (&b)[4], mfma_acc_t acc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
  return acc;
}

// acc += sum_{kb = kb0}^{kb1 - 1} A_kb B_kb (16 x 16 blocks, four k-steps each); load(kb, a, b) fetches the
// operand fragments of block kb from LDS. The fragments of block kb + 1 are requested before the products
// of block kb are issued, so that only the first block pays the LDS round trip.
template <class Load>
__device__ __forceinline__ mfma_acc_t block_chain(mfma_acc_t acc, const int kb0, const int kb1, Load load) {
  if (kb0 >= kb1) return acc;
  double a0[4], b0[4];
  load(kb0, a0, b0);
  for (int kb = kb0; kb < kb1; ++kb) {
    double a1[4], b1[4];
    load(kb + 1 < kb1 ? kb + 1 : kb, a1, b1);
    acc = mfma4(a0, b0, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) { a0[q] = a1[q]; b0[q] = b1[q]; }
  }
  return acc;
}

// Blocked Cholesky of S (lower triangle; different summation grouping than the reference: fast mode
// only): diagonal block + its inverse by one wavefront (chol16_and_inverse), the panel below it and the
// trailing update as rank-16 matrix-core products.
__device__ __forceinline__ void sep_cholesky(const SepGeom& g, double* S, double* Wd, int* __restrict__ info,
                                             const Dims& d, const int b) {
  const int ns = g.ns, tiles = g.tiles, li = g.li, lk = g.lk;
  for (int jb = 0; jb < tiles; ++jb) {
    const int j0 = 16 * jb, rem = tiles - 1 - jb;
    if (g.wave == 0) {
      const bool bad = chol16_and_inverse(S + j0 * ns + j0, ns, Wd + jb * 16 * 17, g.lane);
      if (bad && g.lane == 0) flag_failure(info, d, b);
    }
    __syncthreads();
    const double* Wb = Wd + jb * 16 * 17;
    for (int it = jb + 1 + g.wave; it < tiles; it += g.nwave) {  // L21 = A21 W'
      double a[4], bw[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = S[(16 * it + li) * ns + j0 + 4 * q + lk]; bw[q] = Wb[li * 17 + 4 * q + lk]; }
      const mfma_acc_t acc = mfma4(a, bw, mfma_acc_t{0.0, 0.0, 0.0, 0.0});
      double* Ct = S + (16 * it + lk) * ns + j0 + li;
      Ct[0] = acc[0]; Ct[4 * ns] = acc[1]; Ct[8 * ns] = acc[2]; Ct[12 * ns] = acc[3];
    }
    __syncthreads();
    for (int item = g.wave; item < rem * rem; item += g.nwave) {  // trailing rank-16 update
      const int it = jb + 1 + item / rem, ct = jb + 1 + item % rem;
      if (ct > it) continue;  // lower triangle of tiles only
      double* Ct = S + (16 * it + lk) * ns + 16 * ct + li;
      double a[4], bl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = -S[(16 * it + li) * ns + j0 + 4 * q + lk]; bl[q] = S[(16 * ct + li) * ns + j0 + 4 * q + lk]; }
      mfma_acc_t acc = {Ct[0], Ct[4 * ns], Ct[8 * ns], Ct[12 * ns]};
      acc = mfma4(a, bl, acc);
      Ct[0] = acc[0]; Ct[4 * ns] = acc[1]; Ct[8 * ns] = acc[2]; Ct[12 * ns] = acc[3];
    }
    if (rem > 0) __syncthreads();
  }
}

// W = L^-1, in place over the strictly lower blocks of L (diagonal blocks: Wd), block row by block
// row: T_ij = sum_{k = j}^{i-1} L_ik W_kj,  W_ij = -W_ii T_ij  (j < i). With W the two triangular
// sweeps of every panel chunk become two GEMMs whose tiles are all independent: 4 workgroup
// barriers per chunk instead of 16 (the sweeps made the kernel barrier-bound: eight wavefronts
// share 3..12 tiles per step). One tile of the block row per wavefront (host: nwave >= n / 16).
__device__ __forceinline__ void sep_invert(const SepGeom& g, double* S, const double* Wd) {
  const int ns = g.ns, tiles = g.tiles, li = g.li, lk = g.lk;
  for (int ib = 1; ib < tiles; ++ib) {
    mfma_acc_t wij = {0.0, 0.0, 0.0, 0.0};
    const int jb = g.wave;
    if (jb < ib) {
      const mfma_acc_t t = block_chain(mfma_acc_t{0.0, 0.0, 0.0, 0.0}, jb, ib, [&](const int kb, double (&a)[4], double (&bw)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[q] = S[(16 * ib + li) * ns + 16 * kb + 4 * q + lk];
          bw[q] = kb == jb ? Wd[jb * 16 * 17 + (4 * q + lk) * 17 + li] : S[(16 * kb + 4 * q + lk) * ns + 16 * jb + li];
        }
      });
      // component q of an accumulator is element (4 q + lk, li): exactly the B operand of k-step q
      double wd[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wd[q] = -Wd[ib * 16 * 17 + li * 17 + 4 * q + lk];
#pragma unroll
      for (int q = 0; q < 4; ++q) wij = __builtin_amdgcn_mfma_f64_16x16x4f64(wd[q], t[q], wij, 0, 0, 0);
    }
    __syncthreads();  // every L_ik of the block row has been read
    if (jb < ib) {
      double* Ct = S + (16 * ib + lk) * ns + 16 * jb + li;
      Ct[0] = wij[0]; Ct[4 * ns] = wij[1]; Ct[8 * ns] = wij[2]; Ct[12 * ns] = wij[3];
    }
    __syncthreads();
  }
}

// X <- W' (W X) for the tc column tiles of the panel chunk in LDS: Y = W R (block lower triangular),
// then X = W' Y. Every tile of a product is independent; the wavefronts keep theirs in the
// accumulators across the barrier that separates reading from overwriting the panel. Ends with a
// workgroup barrier (the solved chunk is visible to everybody).
__device__ __forceinline__ void sep_panel_solve(const SepGeom& g, const double* S, const double* Wd, double* X,
                                                const int tc) {
  const int ns = g.ns, xs = g.xs, tiles = g.tiles, li = g.li, lk = g.lk, wave = g.wave, nwave = g.nwave;
  constexpr int MAXI = kSepMaxPanelTiles;
  mfma_acc_t accs[MAXI];
#pragma unroll
  for (int idx = 0; idx < MAXI; ++idx) {
    const int item = wave + idx * nwave;
    mfma_acc_t acc = {0.0, 0.0, 0.0, 0.0};
    if (item < tiles * tc) {
      const int it = item / tc, ct = item % tc;
      // block (it, kb) of W as A operand (row li, k = 4 q + lk)
      acc = block_chain(acc, 0, it + 1, [&](const int kb, double (&a)[4], double (&bx)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[q] = kb == it ? Wd[it * 16 * 17 + li * 17 + 4 * q + lk] : S[(16 * it + li) * ns + 16 * kb + 4 * q + lk];
          bx[q] = X[(16 * kb + 4 * q + lk) * xs + 16 * ct + li];
        }
      });
    }
    accs[idx] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int idx = 0; idx < MAXI; ++idx) {
    const int item = wave + idx * nwave;
    if (item < tiles * tc) {
      const int it = item / tc, ct = item % tc;
      double* Ct = X + (16 * it + lk) * xs + 16 * ct + li;
      Ct[0] = accs[idx][0]; Ct[4 * xs] = accs[idx][1]; Ct[8 * xs] = accs[idx][2]; Ct[12 * xs] = accs[idx][3];
    }
  }
  __syncthreads();
#pragma unroll
  for (int idx = 0; idx < MAXI; ++idx) {
    const int item = wave + idx * nwave;
    mfma_acc_t acc = {0.0, 0.0, 0.0, 0.0};
    if (item < tiles * tc) {
      const int it = item / tc, ct = item % tc;
      // block (it, kb) of W': W'(16 it + li, 16 kb + 4 q + lk) = W(16 kb + 4 q + lk, 16 it + li)
      acc = block_chain(acc, it, tiles, [&](const int kb, double (&a)[4], double (&bx)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a[q] = kb == it ? Wd[it * 16 * 17 + (4 * q + lk) * 17 + li] : S[(16 * kb + 4 * q + lk) * ns + 16 * it + li];
          bx[q] = X[(16 * kb + 4 * q + lk) * xs + 16 * ct + li];
        }
      });
    }
    accs[idx] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int idx = 0; idx < MAXI; ++idx) {
    const int item = wave + idx * nwave;
    if (item < tiles * tc) {
      const int it = item / tc, ct = item % tc;
      double* Ct = X + (16 * it + lk) * xs + 16 * ct + li;
      Ct[0] = accs[idx][0]; Ct[4 * xs] = accs[idx][1]; Ct[8 * xs] = accs[idx][2]; Ct[12 * xs] = accs[idx][3];
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------- separator
// The separator of separator_generic (ndlqr_FactorInnerProduct nested_dissection.c:114-134, the
// Cholesky of src/solve.c:87-98, ndlqr_SolveCholeskyFactor :136-152) for blocks that fill 16x16
// tiles (n a multiple of 16, n + m of 4), fast mode: inner products, a Cholesky blocked by 16
// columns (diagonal block + its inverse by one wavefront, chol16_and_inverse; panel and trailing
// updates as rank-16 products), the blocked inverse W = L^-1 and X = W'(W R) on v_mfma_f64_16x16x4_f64.
// The 2n + 1 right-hand-side columns [f_a | f_bb | z_sep] go through LDS in chunks of CT column
// tiles: S-bar / L (n x (n + 1)), ONE chunk (n x (16 CT + 1)) and the inverses of the diagonal
// blocks are resident -- 67 KB at n = 64 instead of 116 KB for the whole panel, so that two
// workgroups share a CU and the barriers of one overlap with the products of the other.
//   grid (N >> (l+1), batch), block 64 * nwave, dynamic LDS = n (n + 1) + n (16 min(CT, ctl) + 1) + 17 n doubles.
// scratch != nullptr (blocks beyond 112 states, round 4): S-bar / L lives in global memory -- per workgroup
// `scratch_pitch` doubles at scratch + workgroup index * scratch_pitch, in the L2 of its XCD -- and the LDS holds the panel
// chunk and the inverses of the diagonal blocks alone (n (16 min(CT, ctl) + 1) + 17 n doubles: 135 KB at n = 256).
template <int CT>
__global__ void separator_mfma(Dims d, int l, const double* __restrict__ AB, double* F, double* z,
                               int* __restrict__ info, double* __restrict__ rec, double* scratch = nullptr,
                               const size_t scratch_pitch = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = d.n, w = d.w, N = d.N;
  const int b = blockIdx.y;
  const int half = 1 << l, base = blockIdx.x * (2 << l), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const int ns = n + 1, tiles = n >> 4, ctl = 2 * tiles + 1;  // column tiles: f_a, f_bb, [z_sep | padding]
  const int ctc = ctl < CT ? ctl : CT, xs = 16 * ctc + 1;
  double* S = scratch ? scratch + ((size_t)b * gridDim.x + blockIdx.x) * scratch_pitch : sm;
  double* X = scratch ? sm : S + n * ns;
  double* Wd = X + (size_t)n * xs;  // n / 16 blocks of 16 x 17: inverses of the diagonal blocks of L
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwave = blockDim.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const SepGeom geo = {n, ns, xs, tiles, lane, wave, nwave, li, lk};

  const double* ab = AB + ((size_t)b * N + s) * n * w;
  const double* Es = Fblk(F, d, b, l, s);
  const double* Es1 = Fblk(F, d, b, l, s + 1);
  const double* Fas = a >= 0 ? Fblk(F, d, b, a, s) : Es;
  const double* Fbs1 = bb >= 0 ? Fblk(F, d, b, bb, s + 1) : Es1;
  const double* zsl = z + ((size_t)b * N + s) * d.rows;
  double* zs1 = z + ((size_t)b * N + s + 1) * d.rows;
  const int ksteps = w / 4;

  // 16x16 tile (rt, ct) of [A_s | B_s] * (state + input rows of a factor block): operand fragments of ten
  // k-steps at a time, all requested before the first product
  auto product_tile = [&](const double* Bsrc, const int rt, const int ct) -> mfma_acc_t {
    const double* Arow = ab + (size_t)(16 * rt + li) * w + lk;
    const double* Bcol = Bsrc + (size_t)n * n + (size_t)lk * n + 16 * ct + li;  // rows n.. of the block
    mfma_acc_t acc = {0.0, 0.0, 0.0, 0.0};
    constexpr int CH = 10;
    for (int q0 = 0; q0 < ksteps; q0 += CH) {
      double af[CH], bf[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int q = q0 + c < ksteps ? q0 + c : ksteps - 1;
        af[c] = Arow[4 * q];
        bf[c] = Bcol[(size_t)4 * q * n];
      }
#pragma unroll
      for (int c = 0; c < CH; ++c)  // surplus steps of the last chunk multiply by zero
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(q0 + c < ksteps ? af[c] : 0.0, bf[c], acc, 0, 0, 0);
    }
    return acc;
  };

  // ---- S-bar = [A_s | B_s] E(s).xu - E(s+1).x
  for (int item = wave; item < tiles * tiles; item += nwave) {
    const int rt = item / tiles, ct = item % tiles;
    const mfma_acc_t acc = product_tile(Es, rt, ct);
    double* dst = S + (16 * rt + lk) * ns + 16 * ct + li;
    const double* e1 = Es1 + (size_t)(n + 16 * rt + lk) * n + 16 * ct + li;
    double ev[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ev[g] = e1[4 * g * n];
#pragma unroll
    for (int g = 0; g < 4; ++g) dst[4 * g * ns] = acc[g] - ev[g];
  }
  __syncthreads();

  sep_cholesky(geo, S, Wd, info, d, b);

  double* outS = rec ? nullptr : Fblk(F, d, b, l, s + 1);
  double* outa = (!rec && a >= 0) ? Fblk(F, d, b, a, s + 1) : nullptr;
  double* outb = (!rec && bb >= 0) ? Fblk(F, d, b, bb, s + 1) : nullptr;
  double* myrec = rec ? rec + ((size_t)b * d.N + s) * (2 * n * n + n) : nullptr;
  if (outS) {  // the Cholesky factor goes to the lambda rows of knot s+1, column l (KEEP)
    for (int i = wave; i < n; i += nwave)
      for (int c = lane; c < n; c += 64) outS[i * n + c] = S[i * ns + c];
    __syncthreads();
  }

  sep_invert(geo, S, Wd);

  // ---- the panel, CT column tiles at a time: build, X = W' (W R), store
  for (int t0 = 0; t0 < ctl; t0 += ctc) {
    const int tc = ctl - t0 < ctc ? ctl - t0 : ctc;
    // build the chunk's tiles: f_a = [A_s | B_s] Fa(s).xu, f_bb = -Fbb(s+1).x, z column = [A_s | B_s] z(s).xu - z(s+1)
    for (int item = wave; item < tiles * tc; item += nwave) {
      const int rt = item / tc, t = item % tc, gt = t0 + t;
      double* dst = X + (16 * rt + lk) * xs + 16 * t + li;
      if (gt < tiles) {
        const mfma_acc_t acc = product_tile(Fas, rt, gt);
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[4 * g * xs] = acc[g];
      } else if (gt < 2 * tiles) {
        const double* src = Fbs1 + (size_t)(n + 16 * rt + lk) * n + 16 * (gt - tiles) + li;
        double tv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) tv[g] = src[4 * g * n];
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[4 * g * xs] = -tv[g];
      } else {  // [z | padding] tile: the padding columns (column 0 is written below)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (li != 0) dst[4 * g * xs] = 0.0;
      }
    }
    if (t0 + tc == ctl) {  // this chunk holds the right-hand-side column: [A_s | B_s] z(s).xu - z(s+1).lambda - z(s+1).x
      const int zc = 16 * (ctl - 1 - t0);
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double* arow = ab + (size_t)i * w;
        double acc = -zs1[i];  // beta = -1 on the old lambda entry (nested_dissection.c:125)
        const double zlast = zs1[n + i];
        for (int k0 = 0; k0 < w; k0 += 16) {  // operands fetched sixteen pairs at a time
          double av[16], zv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const int k = k0 + u < w ? k0 + u : w - 1;
            av[u] = arow[k];
            zv[u] = zsl[n + k];
          }
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (k0 + u < w) acc = fma(av[u], zv[u], acc);
        }
        X[i * xs + zc] = acc - zlast;
      }
    }
    __syncthreads();
    sep_panel_solve(geo, S, Wd, X, tc);
    // stores of the chunk. With records (fast mode without KEEP) the lambda rows of the factor array
    // are dead data (the boundary Schur pass takes f_a, f_bb from the record): only the record and the
    // rhs entry are written.
    for (int i = wave; i < n; i += nwave) {
      for (int cc = lane; cc < 16 * tc; cc += 64) {
        const int gt = t0 + (cc >> 4), cl = cc & 15;
        const double v = X[i * xs + cc];
        if (gt < tiles) {
          const int c = 16 * gt + cl;
          if (myrec) { if (a >= 0) myrec[i * n + c] = v; }
          else if (outa) outa[i * n + c] = v;
        } else if (gt < 2 * tiles) {
          const int c = 16 * (gt - tiles) + cl;
          if (myrec) { if (bb >= 0) myrec[n * n + i * n + c] = v; }
          else if (outb) outb[i * n + c] = v;
        } else if (cl == 0) {
          zs1[i] = v;
          if (myrec) myrec[2 * n * n + i] = v;
        }
      }
    }
    __syncthreads();  // the next chunk overwrites the panel
  }
}

// ------------------------------------------------------------------------------------- Schur update
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

// The same update for blocks beyond 64 states (round 4: the knot-based path serves every block size, DESIGN.md section 3
// "any block size"), runtime-sized: n a multiple of 16, any number of rows (the last row tile is clamped / masked). No LDS:
// a wavefront takes a 16-row tile of the knot through column tiles four at a time, the operand fragments of E and of f
// straight from global memory (f is shared by the boundary knots of a subtree and stays in L2), eight k-steps per round of
// loads. Semantics of schur_mfma / schur_generic (lambda rows, created blocks, rhs entry) unchanged.
//   grid (N, batch) -- or (2 * (N >> (l+1)), batch) in boundary mode --, block 256.
static __global__ __launch_bounds__(256) void schur_mfma_rt(Dims d, int l, double* F, double* z, int boundary,
                                                            const double* recs = nullptr) {
  const int N = d.N, rows = d.rows, n = d.n, b = blockIdx.y;
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
  const int RT = (rows + 15) / 16, CTN = n / 16, KS = n / 4;
  for (int pass = 0; pass < 2; ++pass) {
    const int col = pass == 0 ? a : bb;
    if (col < 0) continue;  // uniform
    const bool created = pass == 0 ? !left : left;
    const double* f = recs ? recs + ((size_t)b * N + s) * (2 * (size_t)n * n + n) + (pass == 0 ? 0 : (size_t)n * n)
                           : Fblk(F, d, b, col, s + 1);
    double* g = Fblk(F, d, b, col, i);
    for (int t = wave; t < RT; t += 4) {
      const bool lamtile = 16 * t < n;  // n % 16 == 0: a tile is entirely lambda rows or not
      if (lamtile && recs) continue;
      if (lamtile && !calc_lambda) {
        if (created && i != s + 1) {  // explicit zeros, like schur_generic
          for (int e = lane; e < 16 * n; e += 64) g[(size_t)(16 * t) * n + e] = 0.0;
        }
        continue;
      }
      const int ra = 16 * t + li < rows ? 16 * t + li : rows - 1;  // (A operand row of this lane, clamped)
      const double* Erow = E + (size_t)ra * n + lk;
      for (int c0 = 0; c0 < CTN; c0 += 4) {
        mfma_acc_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ct = c0 + j < CTN ? c0 + j : CTN - 1;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int r = 16 * t + lk + 4 * gq;
            acc[j][gq] = (created || r >= rows) ? 0.0 : g[(size_t)r * n + 16 * ct + li];
          }
        }
        for (int q0 = 0; q0 < KS; q0 += 8) {
          double af[8], bf[4][8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int q = q0 + u < KS ? q0 + u : KS - 1;
            af[u] = q0 + u < KS ? -Erow[4 * q] : 0.0;  // g -= E f
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ct = c0 + j < CTN ? c0 + j : CTN - 1;
              bf[j][u] = f[(size_t)(4 * q + lk) * n + 16 * ct + li];
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[j][u], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (c0 + j >= CTN) continue;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int r = 16 * t + lk + 4 * gq;
            if (r < rows) g[(size_t)r * n + 16 * (c0 + j) + li] = acc[j][gq];
          }
        }
      }
    }
  }
  // rhs entry per row: z(i)[r] -= E(r,:) . z_sep
  const double* zsep = z + ((size_t)b * N + s + 1) * rows;
  for (int r = threadIdx.x; r < rows; r += 256) {
    if (r < n && (recs || !calc_lambda)) continue;
    double* zp = z + ((size_t)b * N + i) * rows + r;
    const double* Erow = E + (size_t)r * n;
    double acc = *zp;
    for (int k0 = 0; k0 < n; k0 += 16) {  // sixteen operand pairs per load round, same order of summation
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
