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
// diagonal: leftovers), w = COLUMN i of W = L^-1 (w[k] = W(k, i), zero for k < i). Left-looking, fused
// with the forward substitution of the unit vectors; step j takes row j of L from lane j inside the
// FMAs. Returns true when a pivot was not positive (NaNs propagate to the last pivot).
template <int NX>
__device__ __forceinline__ bool rb_chol_inv(const int i, double (&acc)[NX], double (&w)[NX]) {
  bool bad = false;
  sfor<NX>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    double v = acc[j], sacc = (j == i) ? 1.0 : 0.0;
    sfor<j>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      fnmac_bc<j>(v, acc[k], acc[k]);   // v    -= L(j, k) L(i, k)
      fnmac_bc<j>(sacc, acc[k], w[k]);  // sacc -= L(j, k) W(k, i)
    });
    dpp_fence(v);  // v was written by the instruction before last
    const double pivot = row_bc<j>(v);
    // 1 / sqrt(pivot): hardware estimate + one Newton step (a few ulp; see factor_solve_mc)
    const double y0 = __builtin_amdgcn_rsq(pivot);
    const double e = fma(-pivot * y0, y0, 1.0);
    const double rinv = fma(y0 * e, 0.5, y0);
    if constexpr (j == NX - 1) bad = !((rinv > 0.0) & (rinv < 1.0e300));
    acc[j] = v * rinv;
    w[j] = sacc * rinv;
    asm volatile("" : "+v"(w[j]));
    dpp_fence(acc[j]);  // both are broadcast from this lane by the steps that follow
  });
  return bad;
}

}  // namespace ndlqr
