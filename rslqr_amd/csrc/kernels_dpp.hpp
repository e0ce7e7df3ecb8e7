// kernels_dpp.hpp -- row-broadcast (DPP) building blocks shared by kernels_rowbcast.hpp (one separator
// per 16-lane row) and kernels_bottom_reduced.hpp (Cholesky + inverse of the matrix-core core).
//   v_fmac_f64_dpp acc, x, y row_newbcast:J   acc += (x of lane J of this 16-lane row) * y
// costs one fp64 FMA issue slot on gfx950 (tools/ubench/issue_rates.hip): the broadcast that
// v_readlane + scalar operand needs three instructions for is a source modifier of the FMA.
#pragma once
#include <utility>

#include "kernels_common.hpp"

namespace ndlqr {

// ------------------------------------------------------------------------------------- primitives
// The DPP instructions read other lanes' registers, so every lane of the wavefront has to execute
// them: they are `asm volatile`, which keeps the compiler from sinking them into the lane-predicated
// regions that only store their results (a pure asm whose sole use sits under `if (i < NX)` would
// otherwise be moved there and read inactive lanes).
// acc += (x of lane J of this 16-lane row) * y
template <int J>
__device__ __forceinline__ void fmac_bc(double& acc, const double x, const double y) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
}
// acc -= (x of lane J of this row) * y
template <int J>
__device__ __forceinline__ void fnmac_bc(double& acc, const double x, const double y) {
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(J));
}
// x of lane J of this row
template <int J>
__device__ __forceinline__ double row_bc(const double x) {
  double r;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(J));
  return r;
}

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// Hardware rule (gfx9 family, "VALU writes VGPR -> DPP reads that VGPR": 2 wait states, not
// interlocked): a register that a DPP instruction is about to read from OTHER lanes must not have been
// written by one of the two preceding instructions. The compiler inserts those wait states for its own
// DPP instructions but cannot see into the asm statements here, and it is free to schedule the
// v_mul that produces an operand right in front of the asm that broadcasts it -- which reads stale
// lanes. dpp_fence(x): every element of x is produced before this point (empty asm statements that
// "modify" them: no instruction, only an ordering edge), then two wait states, then the consumers
// (volatile asm statements keep their order).
__device__ __forceinline__ void dpp_fence(double& x) {
  asm volatile("" : "+v"(x));
  asm volatile("s_nop 1");
}
template <int N>
__device__ __forceinline__ void dpp_fence(double (&x)[N]) {
#pragma unroll
  for (int c = 0; c < N; ++c) asm volatile("" : "+v"(x[c]));
  asm volatile("s_nop 1");
}

// ------------------------------------------------------------------------------------- Cholesky + inverse
// Lane i: row i of S-bar in acc. On return acc = row i of the Cholesky factor L (entries above the
// diagonal: leftovers), w = COLUMN i of W = L^-1 (w[k] = W(k, i), zero for k < i). Fused with the forward
// substitution of the unit vectors. Returns true when a pivot was not positive (NaNs propagate to the last pivot).
//
// RIGHT-looking, software-pipelined (round 3): once column j is scaled, every later column c takes its update
//     acc[c] -= L(c, j) L(i, j)      w[c] -= L(c, j) W(j, i)          (L(c, j): lane c's acc[j], inside the FMA)
// independently of the others, and the next pivot only waits for the update of column j + 1. So the pivot chain of
// step j + 1 (broadcast, v_rsq_f64, Newton step: ~10 dependent instructions) is started right behind that one
// update and runs under the remaining 2 (NX - j - 2) updates of step j. The left-looking form it replaces had, per
// step, a chain of j dependent FMAs in front of the pivot chain: the kernels that use this are bound by the
// dependent-instruction latency of a wavefront at 3-4 wavefronts per SIMD, not by issue slots. Every element still
// receives the same products in the same order (k ascending): results are bit-identical to the left-looking form.
// UNIT = false: w comes in as ANY right-hand-side column (one per lane) and leaves as L^-1 times it -- the forward
// substitution of a panel column rides on the factorisation exactly like that of a unit vector.
template <int NX, bool UNIT = true>
__device__ __forceinline__ bool rb_chol_inv(const int i, double (&acc)[NX], double (&w)[NX]) {
  bool bad = false;
  if constexpr (UNIT) {
#pragma unroll
    for (int k = 0; k < NX; ++k) w[k] = (k == i) ? 1.0 : 0.0;
  }
  // 1 / sqrt(pivot): hardware estimate + one Newton step (a few ulp; see factor_solve_mc)
  auto rsqrt_newton = [](const double pivot) {
    const double y0 = __builtin_amdgcn_rsq(pivot);
    const double e = fma(-pivot * y0, y0, 1.0);
    return fma(y0 * e, 0.5, y0);
  };
  dpp_fence(acc[0]);
  double rinv = rsqrt_newton(row_bc<0>(acc[0]));
  sfor<NX>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j == NX - 1) bad = !((rinv > 0.0) & (rinv < 1.0e300));
    acc[j] = acc[j] * rinv;  // L(i, j)
    w[j] = w[j] * rinv;      // W(j, i)
    asm volatile("" : "+v"(w[j]));
    dpp_fence(acc[j]);  // broadcast from other lanes by everything that follows
    if constexpr (j + 1 < NX) {
      fnmac_bc<j + 1>(acc[j + 1], acc[j], acc[j]);
      fnmac_bc<j + 1>(w[j + 1], acc[j], w[j]);
      dpp_fence(acc[j + 1]);  // complete: the pivot of the next step
      const double rnext = rsqrt_newton(row_bc<j + 1>(acc[j + 1]));
      sfor<NX - j - 2>([&](auto cc) {
        constexpr int c = j + 2 + decltype(cc)::value;
        fnmac_bc<c>(acc[c], acc[j], acc[j]);  // acc[c] -= L(c, j) L(i, j)
        fnmac_bc<c>(w[c], acc[j], w[j]);      // w[c]   -= L(c, j) W(j, i)
      });
      rinv = rnext;
    }
  });
  return bad;
}

}  // namespace ndlqr
