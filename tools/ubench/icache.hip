// icache.hip -- gfx950 micro-benchmark: does the length of straight-line code matter? The same 6 144
// v_fmac_f64_dpp per wavefront as ONE unrolled block (48 KB of code) or as a 768-instruction body
// (6 KB) executed 8 times, full chip, 2 wavefronts per SIMD. (rb_bottom's first version was 47 KB of
// straight-line code and ran 4x slower than its instruction count.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/icache.hip -o tools/ubench/icache.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY8                                                                   \
  "v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"      \
  "v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"      \
  "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"      \
  "v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"      \
  "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"      \
  "v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"      \
  "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"      \
  "v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"

template <int MODE>
__global__ __launch_bounds__(64) void bench(double* out, int trips) {
  double a = out[threadIdx.x & 63], b = out[(threadIdx.x + 7) & 63];
  double c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3;
  if constexpr (MODE == 0) {  // 6144 instructions straight-line
    asm volatile(".rept 768\n" BODY8 ".endr\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  } else if constexpr (MODE == 1) {  // 768-instruction body, 8 trips
    for (int t = 0; t < trips; ++t)
      asm volatile(".rept 96\n" BODY8 ".endr\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  } else {  // 2048-instruction body (16 KB), 3 trips
    for (int t = 0; t < trips; ++t)
      asm volatile(".rept 256\n" BODY8 ".endr\n"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
  }
  out[64 + blockIdx.x % 64] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int MODE>
static void run(const char* name, int trips, int instr) {
  double* out;
  (void)hipMalloc(&out, 256 * sizeof(double));
  (void)hipMemset(out, 0, 256 * sizeof(double));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 16384;  // one wavefront each, 16 per SIMD
  for (int lds : {0, 20000, 40000}) {  // dynamic LDS limits the wavefronts per CU: 0 -> 8+/SIMD, 20 KB -> 8 per CU, 40 KB -> 4 per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bench<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    bench<MODE><<<blocks, 64, lds>>>(out, trips);
    (void)hipEventRecord(e0);
    bench<MODE><<<blocks, 64, lds>>>(out, trips);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // SIMD-time per instruction: ms * 1024 SIMDs / (blocks * instr)
    printf("%-46s lds/wave %5d B: %.3f ms, %.2f ns of SIMD time per instruction (4 cycles at 2.1 GHz = 1.9 ns)\n", name, lds, ms,
           ms * 1e6 * 1024 / ((double)blocks * instr));
  }
  (void)hipFree(out);
}

int main() {
  run<0>("6144 fmac_dpp straight-line (48 KB code)", 1, 6144);
  run<1>("768-instruction body x 8 (6 KB code)", 8, 6144);
  run<2>("2048-instruction body x 3 (16 KB code)", 3, 6144);
  return 0;
}
