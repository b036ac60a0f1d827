// K1: batched waveform -> MFCC on the device (gfx950).
//
// Stages (extract_features, extract_features_construct_dataset.py:24-39):
//   1. resample_*_kernel   polyphase restatement of resampy 'kaiser_best' (librosa.load's default
//                          sr=22050): y[up*q + p] = sum_k H[p][k] * x[down*q + n_p - (left-1) + k].
//                          Input windows are staged in LDS with coalesced loads; for the 16 kHz case
//                          (441/320, 128 taps) each thread keeps its phase's 128 taps in VGPRs and
//                          sweeps a run of q-blocks, so H is read once per workgroup.
//   2. stft_mel_kernel     one workgroup per (clip, frame pair): two reflect-padded Hann-windowed real
//                          frames are packed into ONE complex 2048-point FFT (Stockham autosort,
//                          radix 8-8-8-4 butterflies in registers, LDS ping-pong with padded indices),
//                          separated by conjugate symmetry, squared, pushed through the sparse
//                          Slaney mel filter bank (<= 2 filters per bin, CSR runs) and 10 log10.
//   3. dct_kernel          per clip: global max over all frames -> top_db floor, DCT-II (ortho)
//                          128 -> 20, frame axis cut / zero-padded to utterance_length, coefficient-
//                          major layout, optional fused StandardScaler affine.
#include "common.h"
#include "mfcc_tables.h"

namespace lipasr {

using namespace tables;

struct MfccPlan {
  int sr_in = 0, n_samp = 0, batch_max = 0;
  int up = 1, down = 1, taps = 0, left = 0;
  int n_valid = 0, n_y = 0, n_frames = 0;
  bool identity = false;  // sr_in == 22050
  float* d_h = nullptr;   // [up][taps]
  int* d_noff = nullptr;  // [up]
  float* d_hann = nullptr;
  float* d_tw = nullptr;  // float2 [2048]
  int* d_mel_start = nullptr;
  int* d_mel_len = nullptr;
  int* d_mel_off = nullptr;
  float* d_mel_w = nullptr;
  float* d_dct = nullptr;  // [20][128]
  float* d_y = nullptr;    // [batch_max][n_y]
  float* d_db = nullptr;   // [batch_max][n_frames][128]
  float* d_fmax = nullptr; // [batch_max][n_frames]
  // optional per-kernel HIP-event timing of lipasr_mfcc_f32 (bench.py's live roofline measurement)
  std::vector<hipEvent_t> prof_events;  // 4 per call: start, after resample, after stft_mel, after dct
  int prof_cap = 0, prof_n = 0;
};

void mfcc_plan_free(MfccPlan* p) {
  if (!p) return;
  void* ptrs[] = {p->d_h, p->d_noff, p->d_hann, p->d_tw, p->d_mel_start, p->d_mel_len, p->d_mel_off,
                  p->d_mel_w, p->d_dct, p->d_y, p->d_db, p->d_fmax};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  for (hipEvent_t e : p->prof_events) (void)hipEventDestroy(e);
  delete p;
}

// ---------------------------------------------------------------------------------------------
// stage 1: resample
// ---------------------------------------------------------------------------------------------
constexpr int kRsQBlocks = 10;  // q-blocks (of `up` outputs) per workgroup in the register kernel

// 16 kHz -> 22.05 kHz fast path: taps == 128, up <= 448.  One thread per output phase.
__global__ __launch_bounds__(448) void resample_reg128_kernel(const float* __restrict__ x, int n_samp,
                                                               float* __restrict__ y, int n_valid, int n_y, int up,
                                                               int down, int left, const float* __restrict__ H,
                                                               const int* __restrict__ noff) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [kRsQBlocks*down + 128]
  const int u = blockIdx.y;
  const int q0 = blockIdx.x * kRsQBlocks;
  const int tid = threadIdx.x;
  const float* xu = x + (size_t)u * n_samp;
  const int win = kRsQBlocks * down + 128;
  const int base = q0 * down - (left - 1);  // xs[i] = x[base + i]
  for (int i = tid; i < win; i += 448) {
    const int n = base + i;
    xs[i] = (n >= 0 && n < n_samp) ? xu[n] : 0.0f;
  }
  float h[128];
  int np = 0;
  if (tid < up) {
    const float4* hr = reinterpret_cast<const float4*>(H + (size_t)tid * 128);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const float4 v = hr[k];
      h[4 * k] = v.x; h[4 * k + 1] = v.y; h[4 * k + 2] = v.z; h[4 * k + 3] = v.w;
    }
    np = noff[tid];
  }
  __syncthreads();
  if (tid >= up) return;
  float* yu = y + (size_t)u * n_y;
  for (int qq = 0; qq < kRsQBlocks; ++qq) {
    const int t = (q0 + qq) * up + tid;
    if (t >= n_y) break;
    const float* xp = xs + qq * down + np;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 128; ++k) acc = fmaf(h[k], xp[k], acc);
    yu[t] = (t < n_valid) ? acc : 0.0f;
  }
}

// any rational ratio: one workgroup per q-block, taps from global memory
__global__ __launch_bounds__(256) void resample_generic_kernel(const float* __restrict__ x, int n_samp,
                                                                float* __restrict__ y, int n_valid, int n_y, int up,
                                                                int down, int left, int taps,
                                                                const float* __restrict__ H,
                                                                const int* __restrict__ noff) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [down + taps]
  const int u = blockIdx.y, q = blockIdx.x, tid = threadIdx.x;
  const float* xu = x + (size_t)u * n_samp;
  const int win = down + taps;
  const int base = q * down - (left - 1);
  for (int i = tid; i < win; i += 256) {
    const int n = base + i;
    xs[i] = (n >= 0 && n < n_samp) ? xu[n] : 0.0f;
  }
  __syncthreads();
  float* yu = y + (size_t)u * n_y;
  for (int p = tid; p < up; p += 256) {
    const int t = q * up + p;
    if (t >= n_y) continue;
    const float* hr = H + (size_t)p * taps;
    const float* xp = xs + noff[p];
    float acc = 0.0f;
    for (int k = 0; k < taps; ++k) acc = fmaf(hr[k], xp[k], acc);
    yu[t] = (t < n_valid) ? acc : 0.0f;
  }
}

__global__ __launch_bounds__(256) void copy_pad_kernel(const float* __restrict__ x, int n_samp, float* __restrict__ y,
                                                        int n_y) {
  const int u = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_y; i += gridDim.x * 256)
    y[(size_t)u * n_y + i] = i < n_samp ? x[(size_t)u * n_samp + i] : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// stage 2: STFT -> power -> mel -> dB
// ---------------------------------------------------------------------------------------------
constexpr int kFftLds = 2048 + 64;
__device__ __forceinline__ int padi(int i) { return i + (i >> 5); }

// np.pad(y, 1024, mode='reflect') index: position j relative to y[0], any j, n >= 2
__device__ __forceinline__ int reflect_index(int j, int n) {
  const int period = 2 * (n - 1);
  int m = j % period;
  if (m < 0) m += period;
  return m < n ? m : period - m;
}

struct cpx { float re, im; };
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cpx mul_mi(cpx a) { return {a.im, -a.re}; }  // a * (-i)

__device__ __forceinline__ void dft8(cpx (&v)[8]) {
  const float s = 0.70710678118654752440f;
  cpx a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]), a2 = cadd(v[2], v[6]), a3 = mul_mi(csub(v[2], v[6]));
  cpx a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]), a6 = cadd(v[3], v[7]), a7 = mul_mi(csub(v[3], v[7]));
  cpx b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
  cpx b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = csub(a5, a7);
  // w1 = (1 - i)/sqrt2, w2 = -i, w3 = (-1 - i)/sqrt2
  cpx t5 = {(b5.re + b5.im) * s, (b5.im - b5.re) * s};
  cpx t6 = mul_mi(b6);
  cpx t7 = {(b7.im - b7.re) * s, (-b7.re - b7.im) * s};
  v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
  v[1] = cadd(b1, t5); v[5] = csub(b1, t5);
  v[2] = cadd(b2, t6); v[6] = csub(b2, t6);
  v[3] = cadd(b3, t7); v[7] = csub(b3, t7);
}

__device__ __forceinline__ void dft4(cpx (&v)[4]) {
  cpx a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), a3 = mul_mi(csub(v[1], v[3]));
  v[0] = cadd(a0, a2); v[2] = csub(a0, a2); v[1] = cadd(a1, a3); v[3] = csub(a1, a3);
}

__device__ __forceinline__ void butterfly(cpx (&v)[8]) { dft8(v); }
__device__ __forceinline__ void butterfly(cpx (&v)[4]) { dft4(v); }

// one Stockham pass of radix R over 2048 points: butterfly j reads src[j + r*2048/R], multiplies by
// exp(-2 pi i r k/(Ns R)) with k = j mod Ns, writes dst[(j/Ns) Ns R + k + r Ns]
template <int R>
__device__ __forceinline__ void fft_pass(const float* __restrict__ sre, const float* __restrict__ sim,
                                         float* __restrict__ dre, float* __restrict__ dim, int Ns, int j,
                                         const float2* __restrict__ tw) {
  constexpr int NR = 2048 / R;
  cpx v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = padi(j + r * NR);
    v[r] = {sre[i], sim[i]};
  }
  const int k = j & (Ns - 1);
  if (Ns > 1) {
    const int tstep = k * (2048 / (Ns * R));
#pragma unroll
    for (int r = 1; r < R; ++r) {
      const float2 w = tw[(r * tstep) & 2047];
      v[r] = cmul(v[r], cpx{w.x, w.y});
    }
  }
  butterfly(v);
  const int j0 = (j - k) * R + k;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = padi(j0 + r * Ns);
    dre[i] = v[r].re;
    dim[i] = v[r].im;
  }
}

struct StftArgs {
  const float* y;  // [B][n_y]
  int n_y, n_frames;
  const float* hann;
  const float2* tw;
  const int* mel_start;
  const int* mel_len;
  const int* mel_off;
  const float* mel_w;
  float* db;    // [B][n_frames][128]
  float* fmax;  // [B][n_frames]
};

__global__ __launch_bounds__(256) void stft_mel_kernel(StftArgs a) {
  __shared__ float bufA_re[kFftLds], bufA_im[kFftLds], bufB_re[kFftLds], bufB_im[kFftLds];
  __shared__ float wmax[4];
  const int tid = threadIdx.x;
  const int u = blockIdx.y;
  const int f0 = blockIdx.x * 2, f1 = f0 + 1;
  const bool has1 = f1 < a.n_frames;
  const float* yu = a.y + (size_t)u * a.n_y;
  // frame f covers padded positions [512 f, 512 f + 2048) = y positions [512 f - 1024, ...)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int n = tid + 256 * e;
    const float w = a.hann[n];
    const int j0 = f0 * 512 + n - 1024;
    const float s0 = yu[reflect_index(j0, a.n_y)];
    const float s1 = has1 ? yu[reflect_index(j0 + 512, a.n_y)] : 0.0f;
    const int i = padi(n);
    bufA_re[i] = w * s0;
    bufA_im[i] = w * s1;
  }
  __syncthreads();
  fft_pass<8>(bufA_re, bufA_im, bufB_re, bufB_im, 1, tid, a.tw);
  __syncthreads();
  fft_pass<8>(bufB_re, bufB_im, bufA_re, bufA_im, 8, tid, a.tw);
  __syncthreads();
  fft_pass<8>(bufA_re, bufA_im, bufB_re, bufB_im, 64, tid, a.tw);
  __syncthreads();
  fft_pass<4>(bufB_re, bufB_im, bufA_re, bufA_im, 512, tid, a.tw);
  fft_pass<4>(bufB_re, bufB_im, bufA_re, bufA_im, 512, tid + 256, a.tw);
  __syncthreads();
  // Z = FFT(frame0 + i frame1) in bufA.  X0[k] = (Z[k] + conj Z[N-k])/2, X1[k] = (Z[k] - conj Z[N-k])/(2i).
  // power spectra into bufB_re (frame 0) and bufB_im (frame 1), bins 0..1024 (unpadded indices)
  for (int k = tid; k <= 1024; k += 256) {
    const int i = padi(k), i2 = padi((2048 - k) & 2047);
    const float zr = bufA_re[i], zi = bufA_im[i];
    const float wr = bufA_re[i2], wi = -bufA_im[i2];
    const float x0r = 0.5f * (zr + wr), x0i = 0.5f * (zi + wi);
    const float x1r = 0.5f * (zi - wi), x1i = -0.5f * (zr - wr);
    bufB_re[k] = x0r * x0r + x0i * x0i;
    bufB_im[k] = x1r * x1r + x1i * x1i;
  }
  __syncthreads();
  const int sel = tid >> 7, m = tid & 127;
  const float* P = sel ? bufB_im : bufB_re;
  const int st = a.mel_start[m], ln = a.mel_len[m];
  const float* w = a.mel_w + a.mel_off[m];
  float s = 0.0f;
  for (int i = 0; i < ln; ++i) s = fmaf(w[i], P[st + i], s);
  const float dbv = 10.0f * log10f(fmaxf(1e-10f, s));  // librosa.power_to_db(ref=1, amin=1e-10)
  const int f = sel ? f1 : f0;
  const bool valid = f < a.n_frames;
  if (valid) a.db[((size_t)u * a.n_frames + f) * 128 + m] = dbv;
  const float wm = wave_max(dbv);
  if ((tid & 63) == 0) wmax[tid >> 6] = wm;
  __syncthreads();
  if (tid == 0) a.fmax[(size_t)u * a.n_frames + f0] = fmaxf(wmax[0], wmax[1]);
  if (tid == 128 && has1) a.fmax[(size_t)u * a.n_frames + f1] = fmaxf(wmax[2], wmax[3]);
}

// ---------------------------------------------------------------------------------------------
// stage 3: top_db floor, DCT, layout
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dct_kernel(const float* __restrict__ db, const float* __restrict__ frame_max,
                                                   int n_frames, int L, const float* __restrict__ dct,
                                                   const double* __restrict__ aff_mean,
                                                   const double* __restrict__ aff_scale, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[4];
  const int tid = threadIdx.x, u = blockIdx.x;
  const int tu = n_frames < L ? n_frames : L;  // frames that reach the output
  const int tp = tu | 1;                       // odd row stride: conflict-free transposed store
  float* dbs = sm;                             // [128][tp]
  float* ds = sm + 128 * tp;                   // [20][128]
  float mx = -INFINITY;
  for (int t = tid; t < n_frames; t += 256) mx = fmaxf(mx, frame_max[(size_t)u * n_frames + t]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  for (int i = tid; i < kNMfcc * 128; i += 256) ds[i] = dct[i];
  __syncthreads();
  const float thr = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - 80.0f;  // top_db = 80
  for (int i = tid; i < tu * 128; i += 256) {
    const int t = i >> 7, m = i & 127;
    dbs[m * tp + t] = fmaxf(db[((size_t)u * n_frames + t) * 128 + m], thr);
  }
  __syncthreads();
  const int n_out = kNMfcc * L;
  for (int o = tid; o < n_out; o += 256) {
    const int c = o / L, t = o - c * L;
    float v = 0.0f;
    if (t < tu) {
      const float* dr = ds + c * 128;
#pragma unroll 8
      for (int m = 0; m < 128; ++m) v = fmaf(dr[m], dbs[m * tp + t], v);
    }
    if (aff_mean) v = (float)(((double)v - aff_mean[o]) / aff_scale[o]);
    out[(size_t)u * n_out + o] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// A12 audio-domain noise (attacks.py:73-86, 145-183, 222-245), one workgroup per clip
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t ctr, uint32_t hi0, uint32_t hi1, float (&z)[4]) {
  uint32_t o[4];
  Philox::gen(seed, ctr, hi0, hi1, o);
  const float u0 = Philox::u01(o[0]), u1 = Philox::u01(o[1]), u2 = Philox::u01(o[2]), u3 = Philox::u01(o[3]);
  const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
  const float t0 = 6.283185307179586f * u1, t1 = 6.283185307179586f * u3;
  z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0); z[2] = r1 * cosf(t1); z[3] = r1 * sinf(t1);
}

__global__ __launch_bounds__(256) void add_noise_kernel(float* __restrict__ y, int n, int mode, float p0, float p1,
                                                         uint64_t seed) {
  __shared__ double red[4];
  const int u = blockIdx.x, tid = threadIdx.x;
  float* yu = y + (size_t)u * n;
  float sigma = p0;
  if (mode == 2) {
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += (double)yu[i] * (double)yu[i];
    s = wave_sum_d(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const double watts = ((red[0] + red[1]) + (red[2] + red[3])) / (double)n;
    // noise_avg_watts = 10^((10 log10(P) - snr)/10) = P * 10^(-snr/10)
    sigma = (float)sqrt(watts * pow(10.0, -(double)p0 / 10.0));
  }
  for (int i4 = tid; i4 * 4 < n; i4 += 256) {
    float z[4];
    normal4(seed, (uint64_t)i4, (uint32_t)u, 0u, z);
    float s[4] = {sigma, sigma, sigma, sigma};
    if (mode == 1) {
      float q[4];
      normal4(seed, (uint64_t)i4, (uint32_t)u, 1u, q);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = (fabsf(q[e]) < p0) ? 10.0f * p1 : p1;  // sigma1 = 10 alpha on impulses
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i4 * 4 + e;
      if (i < n) yu[i] += s[e] * z[e];
    }
  }
}

template <typename T>
static int upload(T** dptr, const std::vector<T>& v) {
  LP_HIP(hipMalloc(dptr, v.size() * sizeof(T)));
  LP_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return LIPASR_OK;
}

static int launch_resample(const MfccPlan* p, const float* wav, int batch, float* y, hipStream_t st) {
  if (p->identity) {
    hipLaunchKernelGGL(copy_pad_kernel, dim3(32, batch), dim3(256), 0, st, wav, p->n_samp, y, p->n_y);
    LP_LAUNCH_CHECK();
    return LIPASR_OK;
  }
  const int nq = (p->n_y + p->up - 1) / p->up;
  if (p->taps == 128 && p->up <= 448) {
    const int nb = (nq + kRsQBlocks - 1) / kRsQBlocks;
    const size_t lds = (size_t)(kRsQBlocks * p->down + 128) * sizeof(float);
    hipLaunchKernelGGL(resample_reg128_kernel, dim3(nb, batch), dim3(448), lds, st, wav, p->n_samp, y, p->n_valid,
                       p->n_y, p->up, p->down, p->left, p->d_h, p->d_noff);
  } else {
    const size_t lds = (size_t)(p->down + p->taps) * sizeof(float);
    hipLaunchKernelGGL(resample_generic_kernel, dim3(nq, batch), dim3(256), lds, st, wav, p->n_samp, y, p->n_valid,
                       p->n_y, p->up, p->down, p->left, p->taps, p->d_h, p->d_noff);
  }
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

static int launch_from_22k(const MfccPlan* p, const float* y, int batch, int L, const double* am, const double* as,
                           float* out, hipStream_t st, hipEvent_t mid = nullptr) {
  StftArgs a;
  a.y = y; a.n_y = p->n_y; a.n_frames = p->n_frames; a.hann = p->d_hann;
  a.tw = reinterpret_cast<const float2*>(p->d_tw);
  a.mel_start = p->d_mel_start; a.mel_len = p->d_mel_len; a.mel_off = p->d_mel_off; a.mel_w = p->d_mel_w;
  a.db = p->d_db; a.fmax = p->d_fmax;
  hipLaunchKernelGGL(stft_mel_kernel, dim3((p->n_frames + 1) / 2, batch), dim3(256), 0, st, a);
  LP_LAUNCH_CHECK();
  if (mid) LP_HIP(hipEventRecord(mid, st));
  const int tu = p->n_frames < L ? p->n_frames : L;
  const size_t lds = ((size_t)128 * (tu | 1) + kNMfcc * 128) * sizeof(float);
  if (lds > 150 * 1024) {
    set_error("mfcc: utterance_length %d needs %zu bytes of LDS", L, lds);
    return LIPASR_EUNSUPPORTED;
  }
  if (lds > 48 * 1024)
    LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
  hipLaunchKernelGGL(dct_kernel, dim3(batch), dim3(256), lds, st, p->d_db, p->d_fmax, p->n_frames, L, p->d_dct, am, as,
                     out);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

}  // namespace lipasr

using namespace lipasr;

extern "C" {

int lipasr_mfcc_plan(lipasr_handle_t h, int sr_in, int n_samp, int batch_max) {
  LP_CHECK_ARG(h != nullptr, "lipasr_mfcc_plan: null handle");
  LP_CHECK_ARG(sr_in >= 1000 && sr_in <= 384000, "lipasr_mfcc_plan: sr_in=%d outside [1000, 384000]", sr_in);
  LP_CHECK_ARG(n_samp >= 2 && batch_max >= 1, "lipasr_mfcc_plan: n_samp=%d batch_max=%d", n_samp, batch_max);
  DeviceGuard g(h->device);
  if (h->mfcc) { mfcc_plan_free(h->mfcc); h->mfcc = nullptr; }
  MfccPlan* p = new MfccPlan();
  p->sr_in = sr_in; p->n_samp = n_samp; p->batch_max = batch_max;
  p->identity = (sr_in == kSr);
  resampled_lengths(n_samp, sr_in, kSr, &p->n_valid, &p->n_y);
  if (p->n_y < 2) { delete p; set_error("lipasr_mfcc_plan: clip too short after resampling"); return LIPASR_EINVAL; }
  p->n_frames = 1 + p->n_y / kHop;
  int rc = LIPASR_OK;
  if (!p->identity) {
    Polyphase pp = build_polyphase(sr_in, kSr);
    if (pp.up > 4096 || pp.taps > 2048) {
      delete p;
      set_error("lipasr_mfcc_plan: ratio %d/%d needs %d phases x %d taps; unsupported", pp.up, pp.down, pp.up, pp.taps);
      return LIPASR_EUNSUPPORTED;
    }
    p->up = pp.up; p->down = pp.down; p->taps = pp.taps; p->left = pp.left;
    if ((rc = upload(&p->d_h, pp.h)) != LIPASR_OK || (rc = upload(&p->d_noff, pp.n_off)) != LIPASR_OK) {
      mfcc_plan_free(p);
      return rc;
    }
  }
  MelSparse ms = mel_sparse();
  if ((rc = upload(&p->d_hann, hann_periodic())) != LIPASR_OK || (rc = upload(&p->d_tw, twiddles())) != LIPASR_OK ||
      (rc = upload(&p->d_mel_start, ms.start)) != LIPASR_OK || (rc = upload(&p->d_mel_len, ms.len)) != LIPASR_OK ||
      (rc = upload(&p->d_mel_off, ms.off)) != LIPASR_OK || (rc = upload(&p->d_mel_w, ms.w)) != LIPASR_OK ||
      (rc = upload(&p->d_dct, dct_matrix())) != LIPASR_OK) {
    mfcc_plan_free(p);
    return rc;
  }
  const size_t ny = (size_t)batch_max * p->n_y, ndb = (size_t)batch_max * p->n_frames * 128,
               nfm = (size_t)batch_max * p->n_frames;
  if (hipMalloc(&p->d_y, ny * sizeof(float)) != hipSuccess || hipMalloc(&p->d_db, ndb * sizeof(float)) != hipSuccess ||
      hipMalloc(&p->d_fmax, nfm * sizeof(float)) != hipSuccess) {
    mfcc_plan_free(p);
    set_error("lipasr_mfcc_plan: intermediate allocation failed");
    return LIPASR_ENOMEM;
  }
  h->mfcc = p;
  return LIPASR_OK;
}

int lipasr_mfcc_dims(lipasr_handle_t h, int* n_y, int* n_frames) {
  LP_CHECK_ARG(h && n_y && n_frames, "lipasr_mfcc_dims: null argument");
  if (!h->mfcc) { set_error("lipasr_mfcc_dims: call lipasr_mfcc_plan first"); return LIPASR_ESTATE; }
  *n_y = h->mfcc->n_y;
  *n_frames = h->mfcc->n_frames;
  return LIPASR_OK;
}

static int mfcc_check(const char* fn, lipasr_handle_t h, int batch, int L) {
  LP_CHECK_ARG(h != nullptr, "%s: null handle", fn);
  if (!h->mfcc) { set_error("%s: call lipasr_mfcc_plan first", fn); return LIPASR_ESTATE; }
  LP_CHECK_ARG(batch >= 1 && batch <= h->mfcc->batch_max, "%s: batch %d outside [1, %d]", fn, batch, h->mfcc->batch_max);
  LP_CHECK_ARG(L >= 1, "%s: utterance_length=%d", fn, L);
  return LIPASR_OK;
}

int lipasr_resample_f32(lipasr_handle_t h, const float* wav, int batch, float* y, lipasr_stream_t stream) {
  int rc = mfcc_check("lipasr_resample_f32", h, batch, 1);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(wav && y, "lipasr_resample_f32: null argument");
  return launch_resample(h->mfcc, wav, batch, y, S(stream));
}

int lipasr_mfcc_from_22k(lipasr_handle_t h, const float* y, int batch, int n_y, int utterance_length,
                         const double* affine_mean, const double* affine_scale, float* out, lipasr_stream_t stream) {
  int rc = mfcc_check("lipasr_mfcc_from_22k", h, batch, utterance_length);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(y && out, "lipasr_mfcc_from_22k: null argument");
  LP_CHECK_ARG(n_y == h->mfcc->n_y, "lipasr_mfcc_from_22k: n_y=%d but the plan was made for %d", n_y, h->mfcc->n_y);
  LP_CHECK_ARG((affine_mean == nullptr) == (affine_scale == nullptr), "lipasr_mfcc_from_22k: give both affine arrays or neither");
  return launch_from_22k(h->mfcc, y, batch, utterance_length, affine_mean, affine_scale, out, S(stream));
}

int lipasr_mfcc_f32(lipasr_handle_t h, const float* wav, int batch, int utterance_length, const double* affine_mean,
                    const double* affine_scale, float* out, lipasr_stream_t stream) {
  int rc = mfcc_check("lipasr_mfcc_f32", h, batch, utterance_length);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(wav && out, "lipasr_mfcc_f32: null argument");
  LP_CHECK_ARG((affine_mean == nullptr) == (affine_scale == nullptr), "lipasr_mfcc_f32: give both affine arrays or neither");
  MfccPlan* p = h->mfcc;
  hipEvent_t* ev = (p->prof_n < p->prof_cap) ? &p->prof_events[4 * (size_t)p->prof_n] : nullptr;
  if (ev) LP_HIP(hipEventRecord(ev[0], S(stream)));
  rc = launch_resample(p, wav, batch, p->d_y, S(stream));
  if (rc != LIPASR_OK) return rc;
  if (ev) LP_HIP(hipEventRecord(ev[1], S(stream)));
  rc = launch_from_22k(p, p->d_y, batch, utterance_length, affine_mean, affine_scale, out, S(stream), ev ? ev[2] : nullptr);
  if (rc != LIPASR_OK) return rc;
  if (ev) {
    LP_HIP(hipEventRecord(ev[3], S(stream)));
    p->prof_n++;
  }
  return LIPASR_OK;
}

int lipasr_mfcc_profile_begin(lipasr_handle_t h, int max_calls) {
  LP_CHECK_ARG(h != nullptr && max_calls >= 1 && max_calls <= 100000, "lipasr_mfcc_profile_begin: bad argument");
  if (!h->mfcc) { set_error("lipasr_mfcc_profile_begin: call lipasr_mfcc_plan first"); return LIPASR_ESTATE; }
  DeviceGuard g(h->device);
  MfccPlan* p = h->mfcc;
  while ((int)p->prof_events.size() < 4 * max_calls) {
    hipEvent_t e;
    LP_HIP(hipEventCreate(&e));
    p->prof_events.push_back(e);
  }
  p->prof_cap = max_calls;
  p->prof_n = 0;
  return LIPASR_OK;
}

int lipasr_mfcc_profile_end(lipasr_handle_t h, float* avg_ms3, int* n_calls) {
  LP_CHECK_ARG(h && avg_ms3 && n_calls, "lipasr_mfcc_profile_end: null argument");
  if (!h->mfcc) { set_error("lipasr_mfcc_profile_end: call lipasr_mfcc_plan first"); return LIPASR_ESTATE; }
  MfccPlan* p = h->mfcc;
  double acc[3] = {0, 0, 0};
  for (int i = 0; i < p->prof_n; ++i) {
    LP_HIP(hipEventSynchronize(p->prof_events[4 * (size_t)i + 3]));
    for (int k = 0; k < 3; ++k) {
      float ms = 0.0f;
      LP_HIP(hipEventElapsedTime(&ms, p->prof_events[4 * (size_t)i + k], p->prof_events[4 * (size_t)i + k + 1]));
      acc[k] += ms;
    }
  }
  *n_calls = p->prof_n;
  for (int k = 0; k < 3; ++k) avg_ms3[k] = p->prof_n ? (float)(acc[k] / p->prof_n) : 0.0f;
  p->prof_cap = 0;
  p->prof_n = 0;
  return LIPASR_OK;
}

int lipasr_add_noise_f32(lipasr_handle_t h, float* y, int batch, int n, int mode, float p0, float p1, uint64_t seed,
                         lipasr_stream_t stream) {
  LP_CHECK_ARG(h && y, "lipasr_add_noise_f32: null argument");
  LP_CHECK_ARG(batch >= 1 && n >= 1, "lipasr_add_noise_f32: bad shape %dx%d", batch, n);
  LP_CHECK_ARG(mode >= 0 && mode <= 2, "lipasr_add_noise_f32: mode %d", mode);
  hipLaunchKernelGGL(add_noise_kernel, dim3(batch), dim3(256), 0, S(stream), y, n, mode, p0, p1, seed);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

/* Host-only: copies one constant table (as the kernels see it) into `out`; returns the element count
 * (or a negative error).  which: 0 hann[2048], 1 dct[20*128], 2 dense mel[128*1025], 3 polyphase taps
 * [up*taps] for sr_in, 4 polyphase meta {up, down, taps, left} as floats, 5 phase offsets as floats. */
int lipasr_debug_table(int which, int sr_in, float* out, int cap) {
  std::vector<float> v;
  switch (which) {
    case 0: v = hann_periodic(); break;
    case 1: v = dct_matrix(); break;
    case 2: v = mel_dense(); break;
    case 3: case 4: case 5: {
      LP_CHECK_ARG(sr_in >= 1000 && sr_in != kSr, "lipasr_debug_table: sr_in=%d", sr_in);
      Polyphase pp = build_polyphase(sr_in, kSr);
      if (which == 3) v = pp.h;
      else if (which == 4) v = {(float)pp.up, (float)pp.down, (float)pp.taps, (float)pp.left};
      else v.assign(pp.n_off.begin(), pp.n_off.end());
      break;
    }
    default: set_error("lipasr_debug_table: unknown table %d", which); return LIPASR_EINVAL;
  }
  if (out) {
    LP_CHECK_ARG((size_t)cap >= v.size(), "lipasr_debug_table: capacity %d < %zu", cap, v.size());
    memcpy(out, v.data(), v.size() * sizeof(float));
  }
  return (int)v.size();
}

}  // extern "C"
