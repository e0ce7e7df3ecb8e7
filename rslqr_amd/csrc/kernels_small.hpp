// kernels_small.hpp -- size-specialised kernels for small (nstates, ninputs): one factor-block
// ROW per lane, several knots per 64-wide wavefront, separator right-hand sides staged in LDS.
//
// Same arithmetic (and, with STRICT, the same operation order) as kernels_generic.hpp; see that
// file for the mapping to the reference functions. What changes is the work distribution:
//
//   separator_small  one wavefront per level-l separator s. Lanes are split in groups of NX:
//                    group 0 lane i owns row i of S-bar (and entry i of the rhs), group 1 row i of
//                    the left outer right-hand side f_a, group 2 row i of f_bb. The operands
//                    that every lane needs (state/input rows of knot s) are staged in LDS and
//                    read as broadcasts; A_s, B_s rows stay in registers. The Cholesky runs on
//                    group 0's registers with v_readlane broadcasts of row j; the triangular
//                    solves run one right-hand-side column per lane with L(i,j) as a scalar.
//   schur_small      one wavefront per KPW = 64/ROWS consecutive knots, lane = (knot, row).
//                    Each lane keeps its row of E (column l) in registers and updates its row
//                    of the two live outer columns and its rhs entry; f_a, f_bb, z_sep of the
//                    enclosing subtree are read from LDS as broadcasts (ds_read_b128).
//                    Global traffic is whole rows (NX doubles, 16-byte vector loads/stores).
#pragma once
#include "kernels_common.hpp"

namespace ndlqr {

// value of `v` in lane `src` (src must be wave-uniform): two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

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

// ------------------------------------------------------------------------------------- separator
template <int NX, int NU, bool STRICT>
__global__ __launch_bounds__(64) void separator_small(Dims d, int l, const double* __restrict__ AB,
                                                      double* F, double* z, double* __restrict__ rec,
                                                      int* __restrict__ info) {
  constexpr int W = NX + NU, ROWS = 2 * NX + NU;
  static_assert(3 * NX <= 64, "three lane groups of NX must fit a wavefront");
  __shared__ __attribute__((aligned(16))) double shE[W * NX];   // state+input rows of E(s)
  __shared__ __attribute__((aligned(16))) double shA[W * NX];   // state+input rows of F(s, a)
  __shared__ __attribute__((aligned(16))) double shF[2][NX * NX];  // f_a, f_bb (row-major)
  __shared__ __attribute__((aligned(16))) double shz[W + NX];   // z(s) state+input | z_sep
  const int N = d.N, b = blockIdx.y, lane = threadIdx.x;
  const int half = 1 << l, base = blockIdx.x * (2 << l), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);

  const double* Es = Fblk(F, d, b, l, s);
  const double* Fas = a >= 0 ? Fblk(F, d, b, a, s) : Es;
  const double* zs = z + ((size_t)b * N + s) * ROWS;
  double* zs1 = z + ((size_t)b * N + s + 1) * ROWS;
  for (int e = lane; e < W * NX; e += 64) {
    shE[e] = Es[NX * NX + e];
    shA[e] = Fas[NX * NX + e];
  }
  if (lane < W) shz[lane] = zs[NX + lane];
  __syncthreads();

  const int grp = lane / NX, i = lane - grp * NX;
  // row i of [A_s | B_s]
  double ab[W];
  {
    const double* arow = AB + (((size_t)b * N + s) * NX + i) * W;
#pragma unroll
    for (int k = 0; k < W; ++k) ab[k] = (grp < 2) ? arow[k] : 0.0;
  }
  // ---- P1: row i of S-bar (group 0) / of f_a (group 1): sum_k ab[k] * M[k][:]
  double acc[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) acc[j] = 0.0;
  const double* M = (grp == 1) ? shA : shE;
#pragma unroll
  for (int k = 0; k < W; ++k)
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = mad<STRICT>(ab[k], M[k * NX + j], acc[j]);
  double accz = 0.0;
  if (grp == 0) {
    accz = -zs1[i];  // beta = -1 on the old lambda entry of the rhs
#pragma unroll
    for (int k = 0; k < W; ++k) accz = mad<STRICT>(ab[k], shz[k], accz);
    accz = accz - zs1[NX + i];
    const double* e1 = Fblk(F, d, b, l, s + 1) + (NX + i) * NX;  // state row i of E(s+1)
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = acc[j] - e1[j];
  } else if (grp == 2 && bb >= 0) {
    const double* b1 = Fblk(F, d, b, bb, s + 1) + (NX + i) * NX;  // f_bb = -(state rows of F(s+1,bb))
#pragma unroll
    for (int j = 0; j < NX; ++j) acc[j] = -b1[j];
  }

  // ---- P2: Cholesky of S-bar on group 0's registers (left-looking, column by column).
  // Every lane runs the code (no divergence); only group 0's Lr is meaningful. Entries above
  // the diagonal keep their S-bar values, exactly like the reference's in-place factorisation.
  double Lr[NX];
#pragma unroll
  for (int j = 0; j < NX; ++j) Lr[j] = acc[j];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    if (ok) {
      double v = Lr[j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = mad<STRICT>(-Lr[k], readlane_f64(Lr[k], j), v);
      if (i >= j) Lr[j] = v;
      const double pivot = readlane_f64(Lr[j], j);
      if (!(pivot > 0.0)) {
        ok = false;
      } else {
        const double root = sqrt(pivot);
        if (i >= j) Lr[j] = Lr[j] / root;
      }
    }
  }
  if (!ok && lane == 0) atomicAdd(info + b, 1);

  // ---- hand the right-hand sides over to one-column-per-lane form through LDS
  if (grp == 0) shz[W + i] = accz;
  if (grp == 1 || grp == 2) {
#pragma unroll
    for (int j = 0; j < NX; ++j) shF[grp - 1][i * NX + j] = acc[j];
  }
  __syncthreads();
  // lanes [0,NX): columns of f_a; [NX,2NX): columns of f_bb; lane 2NX: the rhs vector
  const int which = lane / NX, col = lane - which * NX;
  const bool has_col = (which == 0 && a >= 0) || (which == 1 && bb >= 0) || (lane == 2 * NX);
  double x[NX];
#pragma unroll
  for (int k = 0; k < NX; ++k)
    x[k] = (lane == 2 * NX) ? shz[W + k] : (which < 2 ? shF[which][k * NX + col] : 0.0);

  // ---- P3: L y = x, then L' x = y; L(i,j) = register j of lane i in group 0 (scalar broadcast)
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    x[j] = x[j] / readlane_f64(Lr[j], j);
#pragma unroll
    for (int r = j + 1; r < NX; ++r) x[r] = mad<STRICT>(-readlane_f64(Lr[j], r), x[j], x[r]);
  }
#pragma unroll
  for (int j = NX - 1; j >= 0; --j) {
    x[j] = x[j] / readlane_f64(Lr[j], j);
#pragma unroll
    for (int r = 0; r < j; ++r) x[r] = mad<STRICT>(-readlane_f64(Lr[r], j), x[j], x[r]);
  }

  // ---- store: factor of S-bar (rows), f_a / f_bb (columns), z_sep; the three right-hand
  //      sides also go, contiguously, into this separator's record rec[b][s] = f_a | f_bb | z_sep
  //      (what the Schur / apply kernels stage in LDS)
  double* myrec = rec + ((size_t)b * N + s) * (2 * NX * NX + NX);
  if (grp == 0) {
    double* outS = Fblk(F, d, b, l, s + 1) + i * NX;
#pragma unroll
    for (int j = 0; j < NX; ++j) outS[j] = Lr[j];
  }
  if (has_col) {
    if (lane == 2 * NX) {
#pragma unroll
      for (int k = 0; k < NX; ++k) { zs1[k] = x[k]; myrec[2 * NX * NX + k] = x[k]; }
    } else {
      double* out = Fblk(F, d, b, which == 0 ? a : bb, s + 1) + col;
      double* out2 = myrec + which * NX * NX + col;
#pragma unroll
      for (int k = 0; k < NX; ++k) { out[k * NX] = x[k]; out2[k * NX] = x[k]; }
    }
  }
}

// ------------------------------------------------------------------------------------- Schur update
template <int NX, int NU>
struct SchurShape {
  static constexpr int ROWS = 2 * NX + NU;
  static constexpr int KPW = 64 / ROWS;      // knots per wavefront
  static constexpr int WAVES = 4;
  static constexpr int KPB = KPW * WAVES;    // knots per workgroup
  static constexpr int NREC = KPB / 2;       // subtrees a workgroup can span (level 0)
  static constexpr int REC = 2 * NX * NX + NX;  // doubles per record: f_a | f_bb | z_sep
};

// BOUNDARY = false: every knot (grid.x = N / KPB workgroups of KPB consecutive knots).
// BOUNDARY = true : only the first and the last knot of every level-l subtree, the two that
//                   later separators read (one wavefront per subtree, KPW must be 2..; grid.x =
//                   ceil(N / 2^(l+1) / WAVES)); used for the upper levels before apply_small.
template <int NX, int NU, bool STRICT, bool BOUNDARY>
__global__ __launch_bounds__(256) void schur_small(Dims d, int l, double* F, double* z,
                                                   const double* __restrict__ recs) {
  using Sh = SchurShape<NX, NU>;
  constexpr int ROWS = Sh::ROWS, KPW = Sh::KPW, KPB = Sh::KPB, REC = Sh::REC, WAVES = Sh::WAVES;
  static_assert(KPW >= 2 && (KPB & (KPB - 1)) == 0, "knots per workgroup must be a power of two");
  constexpr int NRECS = BOUNDARY ? WAVES : Sh::NREC;
  __shared__ __attribute__((aligned(16))) double rec[NRECS][REC];
  const int N = d.N, b = blockIdx.y;
  const int half = 1 << l, T = 2 << l;
  const int first = BOUNDARY ? 0 : blockIdx.x * KPB;
  const int nrec = BOUNDARY ? WAVES : ((T >= KPB) ? 1 : KPB / T);
  const int nsub = N >> (l + 1);

  // cooperative load of the separator records every knot of this workgroup needs
  for (int q = 0; q < nrec; ++q) {
    int qs;
    if (BOUNDARY) {
      const int sub = blockIdx.x * WAVES + q;
      if (sub >= nsub) break;
      qs = sub * T + half - 1;
    } else {
      qs = (((first + q * T) >> (l + 1)) << (l + 1)) + half - 1;
    }
    const double* src = recs + ((size_t)b * N + qs) * REC;
    for (int e = threadIdx.x; e < REC; e += 256) rec[q][e] = src[e];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kn = lane / ROWS, r = lane - kn * ROWS;
  int i, q;
  if (BOUNDARY) {
    const int sub = blockIdx.x * WAVES + wave;
    if (sub >= nsub || kn >= 2) return;
    i = sub * T + (kn == 0 ? 0 : T - 1);
    q = wave;
  } else {
    if (kn >= KPW) return;
    i = first + wave * KPW + kn;
    q = (T >= KPB) ? 0 : (i - first) / T;
  }
  const int base = (i >> (l + 1)) << (l + 1), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const double* fa = rec[q];
  const double* fb = rec[q] + NX * NX;
  const double* zsep = rec[q] + 2 * NX * NX;
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

  double E[NX];
  load_row<NX>(Fblk(F, d, b, l, i) + r * NX, E);
  if (a >= 0) {
    double* g = Fblk(F, d, b, a, i) + r * NX;
    double acc[NX];
    if (left) {
      load_row<NX>(g, acc);
    } else {
#pragma unroll
      for (int c = 0; c < NX; ++c) acc[c] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int c = 0; c < NX; ++c) acc[c] = mad<STRICT>(-E[k], fa[k * NX + c], acc[c]);
    store_row<NX>(g, acc);
  }
  if (bb >= 0) {
    double* g = Fblk(F, d, b, bb, i) + r * NX;
    double acc[NX];
    if (!left) {
      load_row<NX>(g, acc);
    } else {
#pragma unroll
      for (int c = 0; c < NX; ++c) acc[c] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < NX; ++k)
#pragma unroll
      for (int c = 0; c < NX; ++c) acc[c] = mad<STRICT>(-E[k], fb[k * NX + c], acc[c]);
    store_row<NX>(g, acc);
  }
  {
    double* g = z + ((size_t)b * N + i) * ROWS + r;
    double accz = *g;
#pragma unroll
    for (int k = 0; k < NX; ++k) accz = mad<STRICT>(-E[k], zsep[k], accz);
    *g = accz;
  }
}

// ------------------------------------------------------------------------------------- apply
// All upper levels J..K-1 for every knot in ONE pass (DESIGN.md "boundary-first"): once the
// separator records of those levels exist (separator_small on the boundary knots, which
// schur_small<BOUNDARY> keeps up to date), a knot's updates at successive levels only involve its
// own rows: E (column l), the two live outer columns and its rhs entry stay in registers and
// rotate from level to level; only the rhs (and, with KEEP, the finished columns) go back to HBM.
// Same operations in the same order per element as running schur_small level by level.
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
  for (int l = J; l < K; ++l) {
    const int qs = ((first >> (l + 1)) << (l + 1)) + (1 << l) - 1;
    const double* src = recs + ((size_t)b * N + qs) * REC;
    double* dst = lds + (l - J) * REC;
    for (int e = threadIdx.x; e < REC; e += 256) dst[e] = src[e];
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

    if (active) {
      if (a >= 0) {
        double acc[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) acc[c] = left ? Ca[c] : 0.0;
#pragma unroll
        for (int k = 0; k < NX; ++k)
#pragma unroll
          for (int c = 0; c < NX; ++c) acc[c] = mad<STRICT>(-E[k], fa[k * NX + c], acc[c]);
#pragma unroll
        for (int c = 0; c < NX; ++c) Ca[c] = acc[c];
      }
      if (bb >= 0) {
        double acc[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) acc[c] = left ? 0.0 : Cb[c];
#pragma unroll
        for (int k = 0; k < NX; ++k)
#pragma unroll
          for (int c = 0; c < NX; ++c) acc[c] = mad<STRICT>(-E[k], fb[k * NX + c], acc[c]);
#pragma unroll
        for (int c = 0; c < NX; ++c) Cb[c] = acc[c];
      }
#pragma unroll
      for (int k = 0; k < NX; ++k) zz = mad<STRICT>(-E[k], zsep[k], zz);
    } else if (i == s + 1) {  // lambda rows of knot s+1 receive the separator's results
#pragma unroll
      for (int c = 0; c < NX; ++c) {
        if (a >= 0) Ca[c] = fa[r * NX + c];
        if (bb >= 0) Cb[c] = fb[r * NX + c];
      }
      zz = zsep[r];
    } else {  // lambda rows not yet eliminated: the created column starts as zero
#pragma unroll
      for (int c = 0; c < NX; ++c) { if (left) Cb[c] = 0.0; else Ca[c] = 0.0; }
    }

    // rotate into the roles of level l+1: a left child's right outer column is column l+1
    const bool left_child = (base & T) == 0;
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      if (left_child) { E[c] = Cb[c]; Cb[c] = 0.0; }
      else            { E[c] = Ca[c]; Ca[c] = 0.0; }
    }
  }
  *zp = zz;
}

}  // namespace ndlqr
