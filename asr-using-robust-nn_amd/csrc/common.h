// Internal definitions shared by the liblipasr translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>
#include "../../include/lipasr.h"

namespace lipasr {

void set_error(const char* fmt, ...);

#define LP_CHECK_ARG(cond, ...)                    \
  do {                                             \
    if (!(cond)) {                                 \
      lipasr::set_error(__VA_ARGS__);              \
      return LIPASR_EINVAL;                        \
    }                                              \
  } while (0)

#define LP_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      lipasr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return LIPASR_EHIP;                                                              \
    }                                                                                  \
  } while (0)

// after a kernel launch: surfaces launch-configuration errors without synchronising
#define LP_LAUNCH_CHECK() LP_HIP(hipGetLastError())

struct MfccPlan;

}  // namespace lipasr

struct lipasr_mlp;

struct lipasr_ctx {
  int device = 0;
  // scratch for the spectral kernels (chain products, partial sums, scale factors)
  int rs_target_wgs = 256;  // persistent resampler: workgroups to aim for (lipasr_debug_set key 1); survives re-planning
  float* scratch = nullptr;
  size_t scratch_floats = 0;
  float* zeros = nullptr;  // 256 bytes that stay zero (the LDS-DMA ring kernel's source for k >= K)
  std::vector<hipEvent_t> timers;  // pairs: 2*id = start, 2*id+1 = stop
  std::vector<hipGraphExec_t> graphs;
  std::vector<hipStream_t> streams;        // CU-masked streams made by lipasr_stream_create_masked and still alive
  std::vector<struct lipasr_mlp*> mlps;    // classifier plans made on this handle and still alive
  lipasr::MfccPlan* mfcc = nullptr;               // the handle's default MFCC plan (lipasr_mfcc_plan); also in mfcc_plans
  std::vector<lipasr::MfccPlan*> mfcc_plans;      // every MFCC plan made on this handle and still alive
};

namespace lipasr {

inline hipStream_t S(lipasr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    if (prev == dev) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

constexpr size_t kScratchFloats = 1u << 20;  // 4 MiB

// ---- wave / block reductions (64-lane wavefronts) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Philox4x32-10 (Salmon et al. 2011), counter-based RNG for dropout masks and audio noise.
struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
  }
  __host__ __device__ static inline void gen(uint64_t seed, uint64_t ctr_lo, uint32_t ctr_hi0, uint32_t ctr_hi1,
                                             uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), ctr_hi0, ctr_hi1};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) { round(c, k0, k1); k0 += W0; k1 += W1; }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
  // uniform in (0,1]
  __host__ __device__ static inline float u01(uint32_t x) { return ((x >> 8) + 1u) * (1.0f / 16777216.0f); }
};

// spectral.hip: lipasr_project_product with an optional device counter bumped by its single-workgroup kernel
int project_product_bump(lipasr_handle_t h, float* const* Ws, const int* rows, const int* cols, int n_layers, float rho,
                         const int* order, int n_order, float* norms_out, int* bump, lipasr_stream_t stream);

}  // namespace lipasr
