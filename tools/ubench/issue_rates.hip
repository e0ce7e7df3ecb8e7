// issue_rates.hip -- gfx950 micro-benchmark: issue cost (shader cycles per wave-instruction, one
// SIMD) of the fp64 building blocks the separator core can be made of. Decides between the
// candidates of DESIGN.md section 6 (row-broadcast DPP, 4x4x4 matrix-core blocks, v_readlane).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_rates.hip -o gpurun_out/issue_rates && gpurun_out/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 64   // instructions per accumulator set and loop trip
#define TRIPS 64

template <int MODE>
__global__ void bench(double* out, long long* cyc, int trips) {
  double a = out[threadIdx.x & 63], b = out[(threadIdx.x + 7) & 63];
  double c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
  d4 m0 = {a, b, a, b}, m1 = m0, m2 = m0, m3 = m0;
  __syncthreads();
  const long long t0 = clock64();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if constexpr (MODE == 0) {  // plain v_fma_f64, 8 independent chains
        asm volatile(
            "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
            "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
      } else if constexpr (MODE == 1) {  // v_fmac_f64_dpp row_newbcast (broadcast fused into the FMA)
        asm volatile(
            "v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
      } else if constexpr (MODE == 2) {  // v_mov_b64_dpp + v_fma_f64 pairs (4 pairs = 8 instructions)
        double e0, e1, e2, e3;
        asm volatile(
            "v_mov_b64_dpp %4, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %5, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_mov_b64_dpp %7, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fma_f64 %0, %4, %9, %0\n v_fma_f64 %1, %5, %9, %1\n v_fma_f64 %2, %6, %9, %2\n v_fma_f64 %3, %7, %9, %3\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(a), "v"(b));
      } else if constexpr (MODE == 3) {  // v_readlane x2 + v_fma_f64 with the scalar pair (the round-1 broadcast)
        asm volatile(
            "v_readlane_b32 s20, %8, 3\n v_readlane_b32 s21, %9, 3\n s_nop 0\n v_fma_f64 %0, s[20:21], %10, %0\n"
            "v_readlane_b32 s22, %8, 4\n v_readlane_b32 s23, %9, 4\n s_nop 0\n v_fma_f64 %1, s[22:23], %10, %1\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
            : "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(b) : "s20", "s21", "s22", "s23");
      } else if constexpr (MODE == 4) {  // v_mfma_f64_4x4x4_4b, 4 independent accumulators (8 per r)
        asm volatile(
            "v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %4, %5, %1\n"
            "v_mfma_f64_4x4x4_4b_f64 %2, %4, %5, %2\n v_mfma_f64_4x4x4_4b_f64 %3, %4, %5, %3\n"
            "v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %4, %5, %1\n"
            "v_mfma_f64_4x4x4_4b_f64 %2, %4, %5, %2\n v_mfma_f64_4x4x4_4b_f64 %3, %4, %5, %3\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
      } else if constexpr (MODE == 5) {  // v_mfma_f64_4x4x4_4b, ONE dependent chain
        asm volatile(
            "v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n"
            "v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n"
            "v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n"
            "v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n"
            : "+v"(c0) : "v"(a), "v"(b));
      } else if constexpr (MODE == 6) {  // v_mfma_f64_16x16x4, 4 independent accumulators
        asm volatile(
            "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n"
            "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"
            "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n"
            "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"
            : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(a), "v"(b));
      } else if constexpr (MODE == 7) {  // 4x4x4 MFMA interleaved 1:1 with independent v_fma_f64 (do they overlap?)
        asm volatile(
            "v_mfma_f64_4x4x4_4b_f64 %0, %8, %9, %0\n v_fma_f64 %4, %8, %9, %4\n"
            "v_mfma_f64_4x4x4_4b_f64 %1, %8, %9, %1\n v_fma_f64 %5, %8, %9, %5\n"
            "v_mfma_f64_4x4x4_4b_f64 %2, %8, %9, %2\n v_fma_f64 %6, %8, %9, %6\n"
            "v_mfma_f64_4x4x4_4b_f64 %3, %8, %9, %3\n v_fma_f64 %7, %8, %9, %7\n"
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
      } else if constexpr (MODE == 8) {  // dependent v_fma_f64 chain (latency)
        asm volatile(
            "v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n"
            "v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %0, %1, %2, %0\n"
            : "+v"(c0) : "v"(a), "v"(b));
      } else if constexpr (MODE == 9) {  // dependent v_fmac_f64_dpp chain: acc feeds the broadcast source (the Cholesky recurrence)
        asm volatile(
            "v_fmac_f64_dpp %0, %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %0, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %0, %1 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
            "v_fmac_f64_dpp %0, %0, %1 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
            : "+v"(c0) : "v"(b));
      }
    }
  }
  const long long t1 = clock64();
  double s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + m0[0] + m1[1] + m2[2] + m3[3];
  out[64 + (threadIdx.x & 63)] = s;  // keep everything alive
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int instr_per_rep8) {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 128 * sizeof(double)); (void)hipMalloc(&cyc, 64 * sizeof(long long));
  std::vector<double> h(128, 0.0); for (int i = 0; i < 128; ++i) h[i] = 1e-3 * (i + 1);
  for (int waves_per_simd : {1, 2, 4}) {
    (void)hipMemcpy(out, h.data(), 128 * sizeof(double), hipMemcpyHostToDevice);
    const int threads = 256 * waves_per_simd;  // one workgroup on one CU: 4 SIMDs x waves_per_simd
    bench<MODE><<<1, threads>>>(out, cyc, TRIPS);
    bench<MODE><<<1, threads>>>(out, cyc, TRIPS);
    (void)hipDeviceSynchronize();
    std::vector<long long> c(threads / 64);
    (void)hipMemcpy(c.data(), cyc, c.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double n_instr = (double)TRIPS * (REP / 8) * instr_per_rep8;
    // cycles of SIMD time per wave-instruction = wave cycles / instructions / waves sharing the SIMD
    printf("%-52s waves/SIMD %d: %7.2f cyc per instr per wave, %6.2f cyc of SIMD time per instr\n", name, waves_per_simd,
           c[c.size() / 2] / n_instr, c[c.size() / 2] / n_instr / waves_per_simd);
  }
  (void)hipFree(out); (void)hipFree(cyc);
}

// Two wavefronts per SIMD with DIFFERENT streams: wavefronts 0..3 (one per SIMD) issue `nm` v_mfma_f64_16x16x4
// (4 accumulators), wavefronts 4..7 `nv` v_fma_f64 (8 chains). Do the two streams overlap on a SIMD, or does the
// SIMD execute one at a time (elapsed = sum)?
__global__ void mixed(double* out, long long* cyc, int trips_m, int trips_v) {
  double a = out[threadIdx.x & 63], b = out[(threadIdx.x + 7) & 63];
  double c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
  d4 m0 = {a, b, a, b}, m1 = m0, m2 = m0, m3 = m0;
  const bool mf = (threadIdx.x >> 6) < 4;  // (uniform per wavefront)
  __syncthreads();
  const long long t0 = clock64();
  if (mf) {
    for (int t = 0; t < trips_m; ++t)
      asm volatile(
          "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n"
          "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"
          "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n"
          "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n"
          : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(a), "v"(b));
  } else {
    for (int t = 0; t < trips_v; ++t)
      asm volatile(
          "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
          "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  }
  const long long t1 = clock64();
  out[64 + (threadIdx.x & 63)] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + m0[0] + m1[1] + m2[2] + m3[3];
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

static void run_mixed() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 128 * sizeof(double)); (void)hipMalloc(&cyc, 64 * sizeof(long long));
  std::vector<double> h(128, 0.0); for (int i = 0; i < 128; ++i) h[i] = 1e-3 * (i + 1);
  // 512 x 8 matrix-core instructions (~ 262 k cycles alone) beside 0 / 6144 x 8 vector FMAs (~ 262 k cycles alone) and v.v.
  const int cases[3][2] = {{512, 0}, {0, 6144}, {512, 6144}};
  for (auto& cs : cases) {
    (void)hipMemcpy(out, h.data(), 128 * sizeof(double), hipMemcpyHostToDevice);
    mixed<<<1, 512>>>(out, cyc, cs[0], cs[1]);
    mixed<<<1, 512>>>(out, cyc, cs[0], cs[1]);
    (void)hipDeviceSynchronize();
    long long c[8];
    (void)hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("two wavefronts per SIMD, %4d x 8 v_mfma_f64_16x16x4 on one, %4d x 8 v_fma_f64 on the other: matrix-core wavefront %8lld cycles, vector wavefront %8lld cycles\n",
           cs[0], cs[1], c[0], c[4]);
  }
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run_mixed();
  run<0>("v_fma_f64 (8 chains)", 8);
  run<8>("v_fma_f64 (1 dependent chain)", 8);
  run<1>("v_fmac_f64_dpp row_newbcast (8 chains)", 8);
  run<9>("v_fmac_f64_dpp row_newbcast (dependent, src = acc)", 8);
  run<2>("v_mov_b64_dpp + v_fma_f64 (per pair)", 4);
  run<3>("2 v_readlane + v_fma_f64 sgpr (per triple)", 2);
  run<4>("v_mfma_f64_4x4x4_4b (4 accumulators)", 8);
  run<5>("v_mfma_f64_4x4x4_4b (1 dependent chain)", 8);
  run<6>("v_mfma_f64_16x16x4 (4 accumulators)", 8);
  run<7>("v_mfma_f64_4x4x4_4b + v_fma_f64 interleaved (per pair)", 4);
  return 0;
}
