// micro-benchmark: cost of an in-kernel grid barrier (sense-reversing, agent-scope atomics) vs a dependent launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar { unsigned cnt; unsigned gen; unsigned fail; unsigned pad; };

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned n_blocks) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned my_gen = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();  // release this workgroup's writes
    const unsigned arrived = __hip_atomic_fetch_add(&b->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == n_blocks - 1) {
      __hip_atomic_store(&b->cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 20000000u) { ok = false; __hip_atomic_store(&b->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __threadfence();  // acquire
  }
  __syncthreads();
  return ok;
}

__global__ void bar_kernel(Bar* b, int n_bar, float* data, int n_per_block) {
  float acc = 0.f;
  for (int k = 0; k < n_bar; ++k) {
    // a little work touching memory written by another block in the previous phase
    const int peer = (blockIdx.x + k + 1) % gridDim.x;
    acc += data[(size_t)peer * n_per_block + threadIdx.x];
    data[(size_t)blockIdx.x * n_per_block + threadIdx.x] = acc + 1.0f;
    if (!grid_barrier(b, gridDim.x)) return;
  }
}

__global__ void chain_kernel(float* data, int n_per_block, int k) {
  const int peer = (blockIdx.x + k + 1) % gridDim.x;
  const float v = data[(size_t)peer * n_per_block + threadIdx.x];
  data[(size_t)blockIdx.x * n_per_block + threadIdx.x] = v + 1.0f;
}

int main() {
  Bar* b; float* data;
  const int maxb = 1024, npb = 512;
  CK(hipMalloc(&b, sizeof(Bar))); CK(hipMemset(b, 0, sizeof(Bar)));
  CK(hipMalloc(&data, (size_t)maxb * npb * sizeof(float))); CK(hipMemset(data, 0, (size_t)maxb * npb * sizeof(float)));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n_bar = 50;
  for (int blocks : {32, 128, 256, 512}) {
    for (int threads : {256, 512}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(bar_kernel, dim3(blocks), dim3(threads), 0, st, b, n_bar, data, npb);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        Bar hb; CK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost));
        if (rep == 1) printf("barrier kernel: %4d blocks x %3d threads, %d barriers: %8.1f us total, %6.2f us per barrier (fail=%u)\n", blocks, threads, n_bar, ms * 1e3, ms * 1e3 / n_bar, hb.fail);
        if (hb.fail) return 2;
      }
      // the same phases as dependent launches, replayed from a graph
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < n_bar; ++k) hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(threads), 0, st, data, npb, k);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1) printf("graph of launches: %4d blocks x %3d threads, %d kernels:  %8.1f us total, %6.2f us per kernel\n", blocks, threads, n_bar, ms * 1e3, ms * 1e3 / n_bar);
      }
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
