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
                                                      double* F, double* z, int* __restrict__ info) {
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

  // ---- store: factor of S-bar (rows), f_a / f_bb (columns), z_sep
  if (grp == 0) {
    double* outS = Fblk(F, d, b, l, s + 1) + i * NX;
#pragma unroll
    for (int j = 0; j < NX; ++j) outS[j] = Lr[j];
  }
  if (has_col) {
    if (lane == 2 * NX) {
#pragma unroll
      for (int k = 0; k < NX; ++k) zs1[k] = x[k];
    } else {
      double* out = Fblk(F, d, b, which == 0 ? a : bb, s + 1) + col;
#pragma unroll
      for (int k = 0; k < NX; ++k) out[k * NX] = x[k];
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

template <int NX, int NU, bool STRICT>
__global__ __launch_bounds__(256) void schur_small(Dims d, int l, double* F, double* z) {
  using Sh = SchurShape<NX, NU>;
  constexpr int ROWS = Sh::ROWS, KPW = Sh::KPW, KPB = Sh::KPB, REC = Sh::REC;
  static_assert(KPW >= 1 && (KPB & (KPB - 1)) == 0, "knots per workgroup must be a power of two");
  __shared__ __attribute__((aligned(16))) double rec[Sh::NREC][REC];
  const int N = d.N, b = blockIdx.y;
  const int first = blockIdx.x * KPB;
  const int half = 1 << l, T = 2 << l;
  const int nrec = (T >= KPB) ? 1 : KPB / T;

  // cooperative load of the separator results every knot of this workgroup needs
  for (int q = 0; q < nrec; ++q) {
    const int qbase = ((first + q * T) >> (l + 1)) << (l + 1);
    const int qs = qbase + half - 1;
    int qa, qb;
    outer_columns(qbase, l, N, qa, qb);
    const double* fa = qa >= 0 ? Fblk(F, d, b, qa, qs + 1) : nullptr;
    const double* fb = qb >= 0 ? Fblk(F, d, b, qb, qs + 1) : nullptr;
    const double* zsep = z + ((size_t)b * N + qs + 1) * ROWS;
    for (int e = threadIdx.x; e < REC; e += 256) {
      double v = 0.0;
      if (e < NX * NX) { if (fa) v = fa[e]; }
      else if (e < 2 * NX * NX) { if (fb) v = fb[e - NX * NX]; }
      else v = zsep[e - 2 * NX * NX];
      rec[q][e] = v;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kn = lane / ROWS, r = lane - kn * ROWS;
  if (kn >= KPW) return;
  const int i = first + wave * KPW + kn;
  const int base = (i >> (l + 1)) << (l + 1), s = base + half - 1;
  int a, bb;
  outer_columns(base, l, N, a, bb);
  const int q = (T >= KPB) ? 0 : (i - first) / T;
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

}  // namespace ndlqr
