// Developer microbenchmark (GPU box): do H2D copies, D2H copies and kernels on different streams overlap on this
// platform?  hipcc --offload-arch=gfx950 -O2 -o copy_overlap.bin copy_overlap.hip && ./copy_overlap.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(double* p, int iters) {
  double v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
__global__ void kcopy(double2* __restrict__ dst, const double2* __restrict__ src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {  // four 16-byte loads in flight per thread
    const double2 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 59u << 20;
  double *h0, *h1, *d0, *d1, *dk;
  CK(hipHostMalloc((void**)&h0, bytes, hipHostMallocDefault)); CK(hipHostMalloc((void**)&h1, bytes, hipHostMallocDefault));
  CK(hipMalloc(&d0, bytes)); CK(hipMalloc(&d1, bytes)); CK(hipMalloc(&dk, 1 << 24));
  CK(hipMemset(dk, 0, 1 << 24));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  auto h2d = [&](hipStream_t s) { CK(hipMemcpyAsync(d0, h0, bytes, hipMemcpyHostToDevice, s)); };
  auto d2h = [&](hipStream_t s) { CK(hipMemcpyAsync(h1, d1, bytes, hipMemcpyDeviceToHost, s)); };
  auto ker = [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, s, dk, 4000); };
  auto sync = [&]() { CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); };
  for (int w = 0; w < 3; ++w) { h2d(s0); d2h(s1); ker(s0); sync(); }
  double t;
  t = now(); for (int i = 0; i < 5; ++i) h2d(s0); sync(); printf("H2D alone          %.3f ms each\n", (now() - t) / 5);
  t = now(); for (int i = 0; i < 5; ++i) d2h(s1); sync(); printf("D2H alone          %.3f ms each\n", (now() - t) / 5);
  t = now(); for (int i = 0; i < 5; ++i) ker(s0); sync(); printf("kernel alone       %.3f ms each\n", (now() - t) / 5);
  t = now(); for (int i = 0; i < 5; ++i) { h2d(s0); d2h(s1); } sync(); printf("H2D || D2H         %.3f ms per pair\n", (now() - t) / 5);
  t = now(); for (int i = 0; i < 5; ++i) { h2d(s0); ker(s1); } sync(); printf("H2D || kernel      %.3f ms per pair\n", (now() - t) / 5);
  t = now(); for (int i = 0; i < 5; ++i) { d2h(s0); ker(s1); } sync(); printf("D2H || kernel      %.3f ms per pair\n", (now() - t) / 5);
  t = now(); for (int i = 0; i < 5; ++i) { h2d(s0); ker(s0); d2h(s0); h2d(s1); ker(s1); d2h(s1); } sync();
  printf("two pipelined steps (h2d, kernel, d2h per stream)  %.3f ms per step\n", (now() - t) / 10);
  t = now(); for (int i = 0; i < 10; ++i) { hipStream_t s = (i & 1) ? s1 : s0; ker(s); d2h(s); } sync();
  printf("two pipelined steps (kernel, d2h per stream)       %.3f ms per step\n", (now() - t) / 10);
  t = now(); for (int i = 0; i < 10; ++i) { hipStream_t s = (i & 1) ? s1 : s0; h2d(s); ker(s); } sync();
  printf("two pipelined steps (h2d, kernel per stream)       %.3f ms per step\n", (now() - t) / 10);
  {  // dedicated copy streams: all H2D on su, all D2H on sd, kernels on s0 / s1, events in between
    hipStream_t su, sd; CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    hipEvent_t up[2], done[2], down[2];
    for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&up[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&down[i], hipEventDisableTiming)); }
    for (int rep = 0; rep < 2; ++rep) {
      t = now();
      for (int i = 0; i < 10; ++i) {
        const int k = i & 1; hipStream_t sc = k ? s1 : s0;
        if (i >= 2) CK(hipStreamWaitEvent(su, done[k], 0));  // (the kernel of step i - 2 consumed the staging)
        h2d(su); CK(hipEventRecord(up[k], su));
        CK(hipStreamWaitEvent(sc, up[k], 0)); if (i >= 2) CK(hipStreamWaitEvent(sc, down[k], 0));
        ker(sc); CK(hipEventRecord(done[k], sc));
        CK(hipStreamWaitEvent(sd, done[k], 0)); d2h(sd); CK(hipEventRecord(down[k], sd));
      }
      sync(); CK(hipStreamSynchronize(su)); CK(hipStreamSynchronize(sd));
      printf("three-stream pipeline (su: h2d, s0/s1: kernel, sd: d2h)  %.3f ms per step\n", (now() - t) / 10);
    }
  }
  for (int blocks : {256, 1024, 4096}) {  // copies by kernels that access the pinned host memory directly
    const size_t n16 = bytes / 16;
    auto kh2d = [&](hipStream_t s) { hipLaunchKernelGGL(kcopy, dim3(blocks), dim3(256), 0, s, (double2*)d0, (const double2*)h0, n16); };
    auto kd2h = [&](hipStream_t s) { hipLaunchKernelGGL(kcopy, dim3(blocks), dim3(256), 0, s, (double2*)h1, (const double2*)d1, n16); };
    kh2d(s0); kd2h(s1); sync();
    t = now(); for (int i = 0; i < 5; ++i) kh2d(s0); sync(); printf("[%d blocks] kernel H2D alone   %.3f ms each\n", blocks, (now() - t) / 5);
    t = now(); for (int i = 0; i < 5; ++i) kd2h(s1); sync(); printf("[%d blocks] kernel D2H alone   %.3f ms each\n", blocks, (now() - t) / 5);
    t = now(); for (int i = 0; i < 5; ++i) { kh2d(s0); kd2h(s1); } sync(); printf("[%d blocks] kernel H2D || D2H  %.3f ms per pair\n", blocks, (now() - t) / 5);
    t = now(); for (int i = 0; i < 10; ++i) { hipStream_t s = (i & 1) ? s1 : s0; kh2d(s); ker(s); kd2h(s); } sync();
    printf("[%d blocks] two pipelined steps with kernel copies  %.3f ms per step\n", blocks, (now() - t) / 10);
  }
  // kernel through a graph
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal)); ker(s1); CK(hipStreamEndCapture(s1, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  t = now(); for (int i = 0; i < 5; ++i) { h2d(s0); CK(hipGraphLaunch(ge, s1)); } sync(); printf("H2D || graph kernel %.3f ms per pair\n", (now() - t) / 5);
  return 0;
}
