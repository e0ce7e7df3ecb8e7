// Developer tool (GPU box): operand / result layout of v_mfma_f64_4x4x4_4b_f64 from one-hot operands: for every output
// lane the (A lane, B lane) pairs whose product it accumulates.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma44_layout.hip -o tools/ubench/mfma44_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(double* d) {
  const int l = threadIdx.x, p = blockIdx.x, q = blockIdx.y;
  d[((size_t)p * 64 + q) * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(l == p ? 1.0 : 0.0, l == q ? 1.0 : 0.0, 0.0, 0, 0, 0);
}

int main() {
  const size_t n = 64 * 64 * 64;
  std::vector<double> h(n);
  double* dd;
  if (hipMalloc(&dd, n * 8) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(64, 64), dim3(64), 0, 0, dd);
  if (hipMemcpy(h.data(), dd, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int l = 0; l < 64; ++l) {
    printf("D lane %2d <-", l);
    for (int p = 0; p < 64; ++p)
      for (int q = 0; q < 64; ++q)
        if (h[((size_t)p * 64 + q) * 64 + l] != 0.0) printf(" (A%2d,B%2d)", p, q);
    printf("\n");
  }
  return 0;
}
