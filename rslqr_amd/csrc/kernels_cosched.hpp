// kernels_cosched.hpp -- internal: two solves in ONE launch (round 4).
//
// The bottom kernel of the separator-only schedule (bottom_reduced_mc: leaf phase + tree levels 0, 1) is bound by the
// instructions its wavefronts issue -- the one fp64 pipe of every SIMD is busy from its first cycle to its last -- and
// moves its bytes at half the rate the memory system offers; the back-substitution (rb_backsub) is bound by the memory
// system and issues during well under half of its time. Two hardware queues do not interleave the workgroups of two
// such grids (whichever reached the dispatcher first is dispatched to its end: DESIGN.md section 7), so the solve
// pipeline of rounds 2-4 only ever overlapped the tails of its kernels. Here both kinds of workgroup are in ONE grid:
//
//   bottom_backsub_mc: grid (3 N / 16, batch), block 256. Of every three consecutive workgroups the first runs the
//   bottom levels of SIXTEEN knots of the solve that is starting (four wavefronts, one four-knot group each: the body
//   of bottom_reduced_mc, each wavefront on its own quarter of the LDS block), the other two run the back-substitution
//   of eight knots each of the PREVIOUS solve (the body of rb_backsub), whose tree levels finished before this launch
//   and whose records, multipliers and solution live in the other buffer set. The dispatcher therefore keeps a mix of
//   both resident on every CU: while the back-substitution wavefronts wait for their loads the bottom wavefronts have
//   the fp64 pipe, and the bytes of both are requested over the whole launch instead of one after the other.
//
// Registers: the larger of the two bodies (126 for the bottom group at (12,4): four wavefronts per SIMD, as before);
// LDS: the larger of four bottom scratches and one back-substitution block (33 KB at (12,4): four workgroups per CU).
#pragma once
#include "kernels_bottom_reduced.hpp"
#include "kernels_rowbcast.hpp"

namespace ndlqr {

template <int NX, int NU>
union CoschedLds {
  ReducedLds<NX, NU> bottom[4];
  RbBacksubLds<NX, NU> apply;
};

// *_f: buffer set of the solve whose factorisation starts (bottom role); *_s: set of the solve that is being finished
// (back-substitution role). AB, QR: the inputs, shared by both sets.
template <int NX, int NU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void bottom_backsub_mc(
    Dims d, const double* __restrict__ AB, const double* __restrict__ QR, const double* __restrict__ rhs_f, double* red_f,
    double* __restrict__ rec_f, int* __restrict__ info, const double* __restrict__ rhs_s,
    const double* __restrict__ rec_s, const double* __restrict__ ytop_s, double* __restrict__ z_s) {
  __shared__ CoschedLds<NX, NU> lds;
  const int b = blockIdx.y;
  const int g = blockIdx.x / 3, role = blockIdx.x - 3 * g;  // sixteen knots per triple of workgroups
  if (role == 0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    bottom_group_mc<NX, NU, false, false>(d, 16 * g + 4 * wave, b, lane, AB, QR, rhs_f, red_f, rec_f, nullptr, info, 0, 1,
                                          lds.bottom[wave], nullptr, nullptr);
  } else {
    rb_backsub_body<NX, NU>(d, 16 * g + 8 * (role - 1), b, threadIdx.x, AB, QR, rhs_s, rec_s, ytop_s, z_s, lds.apply);
  }
}

}  // namespace ndlqr
