// Does a second stream's kernel start while the first stream replays a graph of short dependent kernels?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void tiny(float* d, int k) { d[blockIdx.x * 64 + threadIdx.x] += 1.0f; }
__global__ void busy(long long* out, long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
__global__ void stamp(long long* out) { if (threadIdx.x == 0) out[0] = wall_clock64(); }
int main() {
  float* d; long long *ob, *oa0, *oa1;
  CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  CK(hipMalloc(&ob, 64)); CK(hipMalloc(&oa0, 64)); CK(hipMalloc(&oa1, 64));
  hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  for (int use_graph = 0; use_graph < 2; ++use_graph) {
    hipGraph_t g; hipGraphExec_t ge;
    if (use_graph) {
      CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
      hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, sa, oa0);
      for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, sa, d, k);
      hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, sa, oa1);
      CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      if (use_graph) CK(hipGraphLaunch(ge, sa));
      else {
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, sa, oa0);
        for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, sa, d, k);
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, sa, oa1);
      }
      hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, sb, ob, 10000LL);  // 100 MHz wall clock: 10000 = 100 us
      CK(hipDeviceSynchronize());
      long long hb[2], a0, a1;
      CK(hipMemcpy(hb, ob, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&a0, oa0, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&a1, oa1, 8, hipMemcpyDeviceToHost));
      if (rep == 2)
        printf("%s: chain of 200 tiny kernels ran %.1f us (t = 0 .. %.1f); the other stream's kernel started at t = %.1f us, ended %.1f\n",
               use_graph ? "graph " : "stream", (a1 - a0) / 100.0, (a1 - a0) / 100.0, (hb[0] - a0) / 100.0, (hb[1] - a0) / 100.0);
    }
  }
  return 0;
}
