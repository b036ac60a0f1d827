// Which XCD does workgroup L of a launch run on?  (scratch probe: unmasked stream and the classifier's CU-masked stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    out[2 * L] = xcc; out[2 * L + 1] = hwid;
  }
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) { lds[threadIdx.x] += 1.0f; }
}
static void run(hipStream_t st, int gx, int gy, size_t ldsb, const char* name) {
  unsigned* d; hipMalloc(&d, gx * gy * 8); hipMemset(d, 0xff, gx * gy * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(512), ldsb, st, d, 2000);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(gx * gy * 2);
  hipMemcpy(h.data(), d, gx * gy * 8, hipMemcpyDeviceToHost);
  printf("%s grid %d x %d lds %zu: xcc of workgroup L (rows of 16)\n", name, gx, gy, ldsb);
  int rr = 0;
  for (int L = 0; L < gx * gy; ++L) { printf("%u", h[2 * L] & 15); if ((h[2*L] & 15) == (unsigned)(L & 7)) ++rr; if (L % 16 == 15) printf("\n"); else printf(" "); }
  printf("\nround-robin matches: %d of %d\n", rr, gx * gy);
  hipFree(d);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int n_cu = p.multiProcessorCount, words = (n_cu + 31) / 32;
  std::vector<uint32_t> mask(words, 0);
  for (int g = 16; g < 32; ++g) for (int b = 8 * g; b < 8 * g + 8; ++b) mask[b / 32] |= 1u << (b % 32);
  hipStream_t sm, s0; hipStreamCreate(&s0);
  if (hipExtStreamCreateWithCUMask(&sm, words, mask.data()) != hipSuccess) { printf("no mask\n"); return 1; }
  run(s0, 16, 16, 68 * 1024, "unmasked");
  run(sm, 16, 16, 68 * 1024, "masked 128");
  run(sm, 16, 8, 68 * 1024, "masked 128");
  run(sm, 105, 1, 96 * 1024, "masked 128 (1-d)");
  return 0;
}
