// kernels_common.hpp -- shared device-side definitions (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

namespace ndlqr {

struct Dims {
  int n, m, N, K, batch;
  int rows;  // 2n + m  rows of a factor block / entries of an rhs block
  int w;     // n + m    row length of the packed [A | B] input
  int fb;    // rows * n doubles per factor block
  int xoff;  // added to blockIdx.x by the kernels of the separator-only schedule: 0, or the first workgroup of this
             // rank's chunk of the horizon (time-axis sharding, launch_time_shard)
};

// acc + a*b. Fast mode: one fused multiply-add. Strict mode: rounded product, then rounded sum --
// the operation order of the reference's scalar loops (src/linalg_custom.c:31-41, 88-132) as
// its default build (-std=c11, no contraction) executes them. The translation unit is compiled
// with -ffp-contract=off, so the compiler never fuses the strict form.
template <bool STRICT>
__device__ __forceinline__ double mad(double a, double b, double acc) {
  if constexpr (STRICT) {
    return acc + a * b;
  } else {
    return fma(a, b, acc);
  }
}

// A Cholesky pivot was not positive: count it for problem b and in the batch total (slot
// info[batch], the only word the host reads back after every solve).
__device__ __forceinline__ void flag_failure(int* info, const Dims& d, int b) {
  atomicAdd(info + b, 1);
  atomicAdd(info + d.batch, 1);
}

// number of trailing one bits = tree level of separator k (src/binary_tree.c:9-37)
__device__ __forceinline__ int trailing_ones(int k) { return __builtin_ctz(~k); }

__device__ __forceinline__ double* Fblk(double* F, const Dims& d, int b, int level, int k) {
  return F + (((size_t)b * d.K + level) * d.N + k) * d.fb;
}
__device__ __forceinline__ const double* Fblk(const double* F, const Dims& d, int b, int level, int k) {
  return F + (((size_t)b * d.K + level) * d.N + k) * d.fb;
}

// Outer columns of the level-`l` subtree starting at knot `base` (SURVEY.md A.2 restated for the
// block-sparse structure, DESIGN.md "live columns"): a = level of the separator left of the
// subtree, bb = level of the separator right of it; -1 when the subtree touches that end.
__device__ __forceinline__ void outer_columns(int base, int l, int N, int& a, int& bb) {
  const int span = 2 << l;
  a = base > 0 ? __builtin_ctz(base) : -1;
  bb = (base + span < N) ? __builtin_ctz(base + span) : -1;
}

// value of `v` in lane `src` (src must be wave-uniform): two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Orders LDS accesses of ONE wavefront: the DS unit executes a wavefront's operations in order,
// so a compiler-level barrier plus draining the LDS counter is enough (no s_barrier).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
}

// Developer instrumentation (tools/segtime.py; never defined in the shipped build): cycles spent
// between consecutive marks, summed over lane 0 of every wavefront, per segment id.
#ifdef NDLQR_SEGTIME
__device__ unsigned long long ndlqr_seg[128];  // [k] cycle sums, [64 + k] sample counts
// one problem in 32 is sampled so that the atomics do not perturb what they measure
#define SEG_INIT() unsigned long long seg_last = __builtin_amdgcn_s_memtime()
#define SEG(k)                                                                         \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    const unsigned long long seg_now = __builtin_amdgcn_s_memtime();                   \
    if (lane == 0 && (blockIdx.y & 31) == 0) {                                                  \
      atomicAdd(&ndlqr_seg[k], seg_now - seg_last);                                    \
      atomicAdd(&ndlqr_seg[64 + (k)], 1ull);                                           \
    }                                                                                  \
    seg_last = seg_now;                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#elif defined(NDLQR_ISA_MARK)  // developer builds: segment boundaries as comments in the assembly (static instruction counts per phase)
#define SEG_INIT() do {} while (0)
#define SEG(k) asm volatile("; SEGMARK %0" ::"n"(k) : "memory")
#else
#define SEG_INIT() do {} while (0)
#define SEG(k) do {} while (0)
#endif

}  // namespace ndlqr
