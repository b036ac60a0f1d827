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
#include "stft.h"
#include <type_traits>

namespace lipasr {

using namespace tables;

struct MfccPlan {
  lipasr_ctx* ctx = nullptr;  // owning handle
  int sr_in = 0, n_samp = 0, batch_max = 0;
  int up = 1, down = 1, taps = 0, left = 0;
  int n_valid = 0, n_y = 0, n_frames = 0;
  int n_fft = kNFft, hop = kHop;
  bool dft = false;          // short-window variant: STFT as an MFMA contraction (dft_mel_kernel)
  int dft_krows = 0, dft_tiles = 0, dft_rpc = 0;
  float* d_dft = nullptr;    // [dft_krows][dft_tiles*64] windowed DFT matrix
  bool identity = false;  // sr_in == 22050
  float* d_h = nullptr;   // [up][taps]
  int* d_noff = nullptr;  // [up]
  float* d_hband = nullptr;  // [n_ptiles][kRsBand][32]: banded taps of 32-phase tiles (MFMA resampler)
  unsigned int* d_hbandh = nullptr;  // [n_ptiles][2 planes][kRhChunks][64 lanes][8 fp16]: the same taps x 8 as fp16 hi / lo fragments
  int* d_lo = nullptr;       // [n_ptiles]: n_off of each tile's first phase
  int n_ptiles = 0;
  int stage_mask = 0;        // debug/profiling: bit0 skip FFT passes, bit1 skip mel, bit2 use the VALU resampler
  int rs_target_wgs = 256;   // persistent resampler: workgroups to aim for (one per CU; fewer leaves CUs to the other stream)
  // fused resample -> STFT kernel (mfcc_fused_kernel): frame groups of a clip of n_samp samples, {q0, f_begin, f_end, 0}
  int* d_groups = nullptr;
  int n_groups = 0;
  bool fused = false;         // the plan CAN run the fused kernel
  bool prefer_fused = false;  // ... and uses it for plain float32 batches too (lipasr_mfcc_plan_set key 2)
  float* d_hann = nullptr;
  float* d_tw = nullptr;  // float2 [2048]
  int* d_mel_start = nullptr;
  int* d_mel_len = nullptr;
  int* d_mel_off = nullptr;
  float* d_mel_w = nullptr;
  float* d_mel_wlo = nullptr;  // [1025] two-filters-per-bin form (mel_pairs)
  float* d_mel_whi = nullptr;
  int* d_mel_pstart = nullptr;  // [128]
  int* d_mel_plen = nullptr;
  float* d_dct = nullptr;  // DCT-II rows in MFMA fragment order: [32 rows (20 real)][k parity][64]
  float* d_y = nullptr;    // [batch_max][n_y]
  float* d_db = nullptr;   // [batch_max][n_frames][128]
  float* d_fmax = nullptr; // [batch_max][n_frames]
  // optional per-kernel HIP-event timing of lipasr_mfcc_f32 (bench.py's live roofline measurement)
  // 5 per extraction: resample start / end, stft_mel start / end, dct end.  The fused entry records all five; the split
  // entries (lipasr_resample_f32 then lipasr_mfcc_from_22k, as the phase-locked pipeline issues them) fill the same slot.
  std::vector<hipEvent_t> prof_events;
  int prof_cap = 0, prof_n = 0;
  bool prof_half = false;  // slot prof_n already holds a resample timing
  // block-DFT STFT on the matrix pipe (stft_bdft.hip): the default 2048/512 path; stage-mask bit 8 (256) selects the Stockham
  // kernel stft_mel2_kernel instead (the parity reference)
  BdftTables bd;
  int bd_seg = 44;  // frames per workgroup (a multiple of 4; 44 = a whole 1-s clip)
  // the kernel's fused top_db + DCT epilogue (lipasr_mfcc_plan_set key 4): off by default -- measured on batches that are not
  // cache-warm it loses to the separate dct_kernel (STFT 139 + 5 us against 120 + 19 us per 1024 clips, round 4)
  bool bd_fuse_dct = false;
};

void mfcc_plan_free(MfccPlan* p) {
  if (!p) return;
  void* ptrs[] = {p->d_hbandh, p->d_groups, p->d_dft, p->d_mel_wlo, p->d_mel_whi, p->d_mel_pstart, p->d_mel_plen, p->d_hband, p->d_lo, p->d_h, p->d_noff, p->d_hann, p->d_tw, p->d_mel_start, p->d_mel_len, p->d_mel_off,
                  p->d_mel_w, p->d_dct, p->d_y, p->d_db, p->d_fmax};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  for (hipEvent_t e : p->prof_events) (void)hipEventDestroy(e);
  bdft_tables_free(&p->bd);
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


// ---------------------------------------------------------------------------------------------
// stage 1, MFMA form (16 kHz / 8 kHz -> 22.05 kHz: up = 441, 128 taps).
//   For a tile of 32 consecutive phases p0..p0+31 every tap reads an input sample in a band of at most
//   152 consecutive samples (offsets n_p0 .. n_p0+151 from the block start - 63), so
//       Y[utterance i][phase j] = sum_kk  X[i][kk] * Hband[kk][j]
//   is a 32 x 32 x 152 GEMM per (32 utterances, q-block, phase tile) on v_mfma_f32_32x32x2_f32 (exact
//   fp32 fma chain; the zero taps of the band add exact zeros).  One workgroup = 32 utterances x 1
//   q-block: the 32 x 448 input samples sit in LDS (row stride 481 = 1 mod 32: the A-operand read
//   `lane i -> row i` is conflict-free; 61.6 kB, so two workgroups share a CU and one's float4 fill
//   overlaps the other's MFMAs), 7 wavefronts take 2 of the 14 phase tiles each, the 76 tap fragments
//   of a tile are loaded up front (coalesced 128 B per half-wave from the L2-resident 272 kB table).
//   Output rows are phases: 128 B contiguous stores per half-wave.
// ---------------------------------------------------------------------------------------------
constexpr int kRsBand = 152, kRsStride = 481, kRsWaves = 7;
typedef float rs_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64 * kRsWaves) void resample_mfma_kernel(const float* __restrict__ x, int n_samp, int batch,
                                                                       float* __restrict__ y, int n_valid, int n_y,
                                                                       int up, int down, int left, int n_ptiles,
                                                                       const float* __restrict__ Hband,
                                                                       const int* __restrict__ lo, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [32][kRsStride]; xs[i][t] = x_i[down*q - 64 + t]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  // XCD-aware map (speed only, see stft_mel_kernel): blocks L, L+8, ... share an L2 and take consecutive q-blocks of
  // one 32-clip tile, whose input windows overlap by 127 of 447 samples
  int ut, q;
  {
    const int nqb = gridDim.x, L = blockIdx.y * gridDim.x + blockIdx.x;
    const int full = (gridDim.y / 8) * 8 * nqb;
    if (L < full) {
      const int chunk = L >> 3;
      ut = (chunk / nqb) * 8 + (L & 7);
      q = chunk % nqb;
    } else {
      ut = blockIdx.y;
      q = blockIdx.x;
    }
  }
  const int u0 = ut * 32;
  const int base = down * q - left;  // = down*q - 64: one sample before the first tap, 16-byte aligned
  constexpr int kVecPerRow = (kRsStride - 1) / 4;  // 120 float4 = 480 floats per row
  const bool vec = ((n_samp & 3) == 0) && ((down & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (!(dbg & 16)) {
    if (vec) {
      for (int f = tid; f < 32 * kVecPerRow; f += 64 * kRsWaves) {
        const int i = f / kVecPerRow, v = f - i * kVecPerRow;
        const int u = u0 + i, n = base + 4 * v;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < batch && n >= 0 && n + 3 < n_samp) val = *reinterpret_cast<const float4*>(x + (size_t)u * n_samp + n);
        float* d = xs + i * kRsStride + 4 * v;
        d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
      }
      if (tid < 32) xs[tid * kRsStride + kRsStride - 1] = 0.0f;
    } else {
      for (int f = tid; f < 32 * kRsStride; f += 64 * kRsWaves) {
        const int i = f / kRsStride, t = f - i * kRsStride;
        const int u = u0 + i, n = base + t;
        xs[f] = (u < batch && n >= 0 && n < n_samp) ? x[(size_t)u * n_samp + n] : 0.0f;
      }
    }
  }
  __syncthreads();
  for (int r = wave; r < n_ptiles; r += kRsWaves) {
    rs_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const float* hb = Hband + (size_t)r * kRsBand * 32 + h * 32 + li;
    const float* xa = xs + li * kRsStride + lo[r] + 1 + h;
    // all 76 tap fragments of this phase tile go to registers first: 76 coalesced loads in flight at once
    float bq[kRsBand / 2];
#pragma unroll
    for (int s = 0; s < kRsBand / 2; ++s) bq[s] = hb[s * 64];
    __builtin_amdgcn_sched_barrier(0);  // keep every load ahead of the MFMA chain (do not sink them back in)
    if (dbg & 8) {
      acc[0] = bq[0] + bq[75] + xa[0];
    } else {
#pragma unroll
      for (int s = 0; s < kRsBand / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * s], bq[s], acc, 0, 0, 0);
    }
    const int p = 32 * r + li;
    const int t = q * up + p;
    if (p < up && t < n_y) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int u = u0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (u < batch) y[(size_t)u * n_y + t] = t < n_valid ? acc[e] : 0.0f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// stage 1, persistent MFMA form.  Same contraction as resample_mfma_kernel, scheduled for the whole chip:
//   * one workgroup per (32-clip tile, q-range): 32 x 8 = 256 workgroups for 1024 clips, one per CU, each walking
//     6-7 consecutive q-blocks (the one-q-block kernel runs 1600 workgroups in 4 rounds on 512 slots, 3.1 rounds of work);
//   * one wavefront per phase tile (14 wavefronts): its 76 tap fragments are loaded ONCE and stay in registers for the
//     whole q-range (they were re-read from L2 for every q-block and clip tile: 436 MB per 1024 clips);
//   * the input window of the next q-block travels global -> registers while the current one feeds the MFMA chain,
//     then registers -> the other LDS buffer (2 x 61.6 kB), one barrier per q-block.
// ---------------------------------------------------------------------------------------------
constexpr int kRpMaxWaves = 16;

__global__ __launch_bounds__(64 * kRpMaxWaves) __attribute__((amdgpu_waves_per_eu(4, 4)))
void resample_persist_kernel(const float* __restrict__ x, int n_samp, int batch, float* __restrict__ y, int n_valid, int n_y,
                             int up, int down, int left, int nq, int n_tiles, const float* __restrict__ Hband,
                             const int* __restrict__ lo) {
  extern __shared__ __attribute__((aligned(16))) float xs2[];  // [2][32][kRsStride]
  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = phase tile
  const int li = lane & 31, h = lane >> 5;
  // 1-D grid of W workgroups over T clip tiles: tile t gets ceil((t+1) W / T) - ceil(t W / T) of them (so W need not be
  // a multiple of T: the grid is sized to the CUs this stream may use), each a contiguous share of the tile's q-blocks
  const int W = gridDim.x, T = n_tiles;
  const int tile = (int)(((long)blockIdx.x * T) / W);
  const int first = (int)(((long)tile * W + T - 1) / T), next = (int)(((long)(tile + 1) * W + T - 1) / T);
  const int n_ranges = next - first, ri = blockIdx.x - first;
  const int q_begin = (int)(((long)ri * nq) / n_ranges), q_end = (int)(((long)(ri + 1) * nq) / n_ranges);
  const int u0 = tile * 32;
  constexpr int kVecPerRow = (kRsStride - 1) / 4;   // 120 float4 = 480 floats per row
  constexpr int kFillMax = 5;                        // float4 per thread per window: 3840 over >= 768 threads
  // tap fragments of this wavefront's phase tile: loaded once
  float bq[kRsBand / 2];
  {
    const float* hb = Hband + (size_t)wave * kRsBand * 32 + h * 32 + li;
#pragma unroll
    for (int s = 0; s < kRsBand / 2; ++s) bq[s] = hb[s * 64];
  }
  const int lo_r = lo[wave];
  float4 stage[kFillMax];
  // (the thread index is made opaque in both helpers so that their per-slot addresses are recomputed where they are
  // used instead of living in 15-20 registers across the MFMA chain -- the fragments need those registers)
  auto fetch = [&](int q) {
    const int base = down * q - left;
    int tq = tid;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int j = 0; j < kFillMax; ++j) {
      const int f = tq + j * nthreads;
      const int i = f / kVecPerRow, v = f - i * kVecPerRow;
      const int u = u0 + i, n = base + 4 * v;
      stage[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < 32 * kVecPerRow && u < batch && n >= 0 && n + 3 < n_samp)
        stage[j] = *reinterpret_cast<const float4*>(x + (size_t)u * n_samp + n);
    }
  };
  auto deposit = [&](float* xs) {
    int tq = tid;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int j = 0; j < kFillMax; ++j) {
      const int f = tq + j * nthreads;
      if (f < 32 * kVecPerRow) {
        const int i = f / kVecPerRow, v = f - i * kVecPerRow;
        float* d = xs + i * kRsStride + 4 * v;
        d[0] = stage[j].x; d[1] = stage[j].y; d[2] = stage[j].z; d[3] = stage[j].w;
      }
    }
    if (tid < 32) xs[tid * kRsStride + kRsStride - 1] = 0.0f;
  };
  if (q_begin < q_end) {
    fetch(q_begin);
    deposit(xs2);
  }
  __syncthreads();
  int cur = 0;
  for (int q = q_begin; q < q_end; ++q) {
    const bool more = q + 1 < q_end;
    if (more) fetch(q + 1);
    const float* xa = xs2 + cur * 32 * kRsStride + li * kRsStride + lo_r + 1 + h;
    rs_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int s = 0; s < kRsBand / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * s], bq[s], acc, 0, 0, 0);
    const int pp = 32 * wave + li;
    const int t = q * up + pp;
    if (pp < up && t < n_y) {
      float* yb = y + (size_t)u0 * n_y;
      int off = 4 * h * n_y + t;
      asm volatile("" : "+v"(off));  // formed here: sixteen hoisted 64-bit row addresses would cost the fragment registers
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2);
        if (u0 + row + 4 * h < batch) yb[off + row * n_y] = t < n_valid ? acc[e] : 0.0f;
      }
    }
    if (more) deposit(xs2 + (cur ^ 1) * 32 * kRsStride);  // the other buffer: nobody reads it during this q-block
    // LDS-only barrier: __syncthreads() would also wait for this q-block's output stores to reach memory
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------
// stage 1, persistent MFMA form on the fp16 matrix instruction (round 3).  Same schedule as resample_persist_kernel; the
// contraction Y[32 clips][32 phases] = X[32][band] . Hband[band][32] runs on v_mfma_f32_32x32x16_f16 with BOTH operands
// split into two fp16 planes, x = x_hi + x_lo, h = h_hi + h_lo (hi = the value rounded to fp16's 11 significant bits, lo =
// the fp16 rounding of the remainder: 22 bits between them), and three of the four cross terms accumulated in fp32:
//       x h  ~=  x_hi h_hi + x_hi h_lo + x_lo h_hi          (dropped: x_lo h_lo <= 2^-22 |x h|)
// Every product of two fp16 numbers is exact in fp32, so the error is the 2^-22 of the two representations and of the
// dropped term: <= 3 x 2.4e-7 x sum |x h| <= 1e-6 per output sample in the worst case, 1e-7 typically -- the fp32 kernel's
// own accumulation error is 6e-8 x sqrt(128).  tests: 2e-6 against the float64 oracle, as for the fp32 kernel, which stays
// the parity reference (stage-mask bit 4).  Cost: 3 matrix instructions of 32 cycles per 16 taps instead of 8 of 64 cycles
// (v_mfma_f32_32x32x2_f32): 5.3x less matrix-pipe time (44 us of the fp32 kernel's 87 us per 1024 clips were MFMA-busy).
// fp16's exponent range is short: the low plane of a value below 2^-14 x 2^11 = 0.125 falls on the subnormal grid (step
// 2^-24) and a quiet passage at -60 dB would come out with a relative error of 1e-4.  So both operands are scaled by powers
// of two before the split -- the samples by 2^11 (full 22-bit precision down to |x| = 6e-5 = -84 dB, an absolute floor of
// 1.5e-11 below that; |x| must stay below 32, audio is in [-1, 1)), the taps by 2^6 -- and the accumulator by 2^-17
// afterwards (all exact).  The band of a phase tile starts at a multiple of 8
// samples (16-byte aligned ds_read_b128 of 8 consecutive fp16) and is padded to 160 = 10 k-steps.
// LDS per window: 32 rows x {hi[480] | lo[480]} fp16 + 16 bytes = 1936 B per row (121 x 16: the 32 rows of a b128 read
// fall on different bank quads), the same 62 kB as the fp32 window, double-buffered.
// ---------------------------------------------------------------------------------------------
typedef _Float16 rs_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 rs_h4 __attribute__((ext_vector_type(4)));
constexpr int kRhK = 160, kRhChunks = kRhK / 16, kRhRowHalfs = 480, kRhRowBytes = 2 * kRhRowHalfs * 2 + 16;
constexpr float kRhTapScale = 64.0f, kRhSigScale = 2048.0f;

// I16: int16 PCM in (float32 otherwise).  RAGGED: nv[] holds a length per clip (else every row is n_samp long and nv is not read).
template <bool I16, bool RAGGED>
__global__ __launch_bounds__(64 * kRpMaxWaves) __attribute__((amdgpu_waves_per_eu(4, 4)))
void resample_persist_h2_kernel(const void* __restrict__ xv, const int* __restrict__ nv, int sr_in, int n_samp, int batch, float* __restrict__ y, int n_valid, int n_y,
                                int up, int down, int left, int nq, int n_tiles, const uint4* __restrict__ HbandH,
                                const int* __restrict__ lo, int n_ptiles_rt, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xh[];  // [2][32][kRhRowBytes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = phase tile
  const int li = lane & 31, h = lane >> 5;
  const int W = gridDim.x, T = n_tiles;
  const int tile = (int)(((long)blockIdx.x * T) / W);
  const int first = (int)(((long)tile * W + T - 1) / T), next = (int)(((long)(tile + 1) * W + T - 1) / T);
  const int n_ranges = next - first, ri = blockIdx.x - first;
  const int q_begin = (int)(((long)ri * nq) / n_ranges), q_end = (int)(((long)(ri + 1) * nq) / n_ranges);
  const int u0 = tile * 32;
  constexpr int kVecPerRow = kRhRowHalfs / 4;  // 120 float4 per row
  constexpr int kFillMax = 4;  // 16 wavefronts fill (3840 float4 over 1024 threads); the first n_ptiles of them also multiply
  const bool mm = wave < n_ptiles_rt;
  // tap fragments of this wavefront's phase tile, both planes: loaded once  [tile][plane][chunk][lane] x 16 bytes
  rs_h8 bh[kRhChunks], bl[kRhChunks];
  if (mm) {
    const uint4* hb = HbandH + ((size_t)wave * 2 * kRhChunks) * 64 + lane;
#pragma unroll
    for (int c = 0; c < kRhChunks; ++c) {
      const uint4 t0 = hb[c * 64], t1 = hb[(kRhChunks + c) * 64];
      bh[c] = __builtin_bit_cast(rs_h8, t0);
      bl[c] = __builtin_bit_cast(rs_h8, t1);
    }
  }
  const int band0 = mm ? ((lo[wave] + 1) & ~7) : 0;  // first sample of the band in the window, a multiple of 8
  // what a fill holds between its loads and its LDS stores: float4, or the four int16 samples as they came
  using stage_t = typename std::conditional<I16, short4, float4>::type;
  stage_t stage[kFillMax];
  // Fill mapping without divisions: a wavefront moves two window rows (16 wavefronts, 32 rows), lane l the 4-sample columns l and
  // l + 64 of each (120 of the 128 exist): 1 KB contiguous per load instruction (fp32), 512 B per LDS store, and the length of
  // the row's clip is wave-uniform -- it stays in a scalar register (this kernel has no vector register to spare: the tap
  // fragments alone take 80 of its 128).
  const int r0 = u0 + 2 * wave;
  const int n_clip0 = __builtin_amdgcn_readfirstlane((r0 < batch) ? (RAGGED ? min(max(nv[r0], 0), n_samp) : n_samp) : 0);
  const int n_clip1 = __builtin_amdgcn_readfirstlane((r0 + 1 < batch) ? (RAGGED ? min(max(nv[r0 + 1], 0), n_samp) : n_samp) : 0);
  // outputs from int(n ratio) on are zeros (fix_length).  With per-clip lengths that is at most ONE sample per clip that anyone
  // reads (ceil(n ratio) - int(n ratio) <= 1) and stft_mel2_kernel, which knows the clip's length, takes it as zero itself
  const int t_lim = RAGGED ? n_y : n_valid;
  auto fetch = [&](int q) {
    const int base = down * q - left;  // a multiple of 4, as n_samp is
    const stage_t* src = static_cast<const stage_t*>(xv) + (((long)r0 * n_samp + base) >> 2) + lane;
#pragma unroll
    for (int j = 0; j < kFillMax; ++j) {
      const int v = lane + 64 * (j & 1), n = base + 4 * v;
      const int n_clip = (j >> 1) ? n_clip1 : n_clip0;
      stage[j] = stage_t{};
      // (n + 3 < n_samp: n and n_samp are multiples of 4; a clip that ends inside the four is cut in deposit)
      if (v < kVecPerRow && n >= 0 && n < n_clip) stage[j] = src[(j >> 1) * (n_samp >> 2) + 64 * (j & 1)];
    }
  };
  auto deposit = [&](unsigned char* xs, int q) {
    unsigned char* drow = xs + 2 * wave * kRhRowBytes + 8 * lane;
    // a clip that ends inside a group of four: what follows in the row is not the clip's.  Cut under a scalar branch (rare; no
    // select on the path below, where every register is taken -- as it is, this code costs the RAGGED instances two spilled
    // tap fragments, reloaded per q-block)
#pragma unroll
    for (int r = 0; RAGGED && r < 2; ++r) {
      const int rel = (r ? n_clip1 : n_clip0) - (down * q - left);  // samples of the clip in this window (wave-uniform)
      if ((rel & 3) != 0 && rel > 0 && rel < 4 * kVecPerRow) {
        const int g = rel >> 2, cr = rel & 3;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if (lane + 64 * jj == g) {
            stage_t& t = stage[2 * r + jj];
            if (cr <= 1) t.y = 0;
            if (cr <= 2) t.z = 0;
            t.w = 0;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kFillMax; ++j) {
      if (lane + 64 * (j & 1) < kVecPerRow) {
        float e[4];
        if constexpr (I16) {
          const short4 sv = stage[j];
          e[0] = (float)sv.x * (kRhSigScale / 32768.0f); e[1] = (float)sv.y * (kRhSigScale / 32768.0f);
          e[2] = (float)sv.z * (kRhSigScale / 32768.0f); e[3] = (float)sv.w * (kRhSigScale / 32768.0f);
        } else {
          const float4 fv = stage[j];
          e[0] = fv.x * kRhSigScale; e[1] = fv.y * kRhSigScale; e[2] = fv.z * kRhSigScale; e[3] = fv.w * kRhSigScale;
        }
        rs_h4 hi, lw;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float sc = fminf(fmaxf(e[c], -65000.0f), 65000.0f);
          const _Float16 a = (_Float16)sc;
          hi[c] = a;
          lw[c] = (_Float16)(sc - (float)a);
        }
        unsigned char* d = drow + (j >> 1) * kRhRowBytes + 512 * (j & 1);
        *reinterpret_cast<rs_h4*>(d) = hi;
        *reinterpret_cast<rs_h4*>(d + 2 * kRhRowHalfs) = lw;
        // a clip that ends inside this group of four: what follows in the row is not the clip's.  Zeroed after the fact, by
        // the lane that wrote it (rare, and no select on the main path)
      }
    }
  };
  if (q_begin < q_end) {
    fetch(q_begin);
    deposit(xh, q_begin);
  }
  __syncthreads();
  int cur = 0;
  for (int q = q_begin; q < q_end; ++q) {
    const bool more = q + 1 < q_end && !(dbg & 4);  // (dbg: profiling switches, results wrong)
    if (more) fetch(q + 1);
    const unsigned char* xa = xh + cur * 32 * kRhRowBytes + li * kRhRowBytes + 2 * (band0 + 8 * h);
    rs_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    if (mm && !(dbg & 1)) {
#pragma unroll
    for (int c = 0; c < kRhChunks; ++c) {
      const rs_h8 ah = *reinterpret_cast<const rs_h8*>(xa + 32 * c);
      const rs_h8 al = *reinterpret_cast<const rs_h8*>(xa + 32 * c + 2 * kRhRowHalfs);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[c], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[c], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[c], acc, 0, 0, 0);
    }
    }
    const int pp = 32 * wave + li;
    const int t = q * up + pp;
    if (mm && pp < up && t < n_y && !(dbg & 2)) {
      float* yb = y + (size_t)u0 * n_y;
      int off = 4 * h * n_y + t;
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2);
        if (u0 + row + 4 * h < batch) yb[off + row * n_y] = t < t_lim ? acc[e] * (1.0f / (kRhTapScale * kRhSigScale)) : 0.0f;
      }
    }
    if (more) deposit(xh + (cur ^ 1) * 32 * kRhRowBytes, q + 1);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    cur ^= 1;
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
// LDS holds complex points as float2; element e lives at e ^ ((e >> 4) & 7) (no padding).  Unit-stride
// accesses (every read, the writes of passes 3 and 4) stay a permutation inside aligned 8-element blocks, i.e.
// conflict-free for ds_read_b64's 32-lane halves, and the stride-8 scatter of pass 1 lands its 16-lane write
// groups on 16 distinct 8-byte slots (pass 2's stride-64 scatter is 2-way).
constexpr int kFftLds = 2048 + 72;  // + room for the four weighted-power arrays laid over one buffer
__device__ __forceinline__ int padi(int i) { return i ^ ((i >> 4) & 7); }

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
// w^r, w = exp(-2 pi i k/(Ns R)), k = j mod Ns, writes dst[(j/Ns) Ns R + k + r Ns].
// Twiddles: w, w^2, w^4 come from the table, the other powers are one complex product away.
// In place on one LDS buffer: every thread reads its inputs, the workgroup meets (sync_between), then writes.
template <int R, int NB>
__device__ __forceinline__ void fft_pass(float2* __restrict__ buf, int Ns, int j0_, const float2* __restrict__ tw,
                                         const cpx* __restrict__ regs = nullptr) {
  constexpr int NR = 2048 / R;
  cpx vv[NB][R];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (regs) {
        vv[b][r] = regs[r];  // pass 1: butterfly j's inputs x[j + r*256] are exactly what thread j loaded
      } else {
        const float2 t = buf[padi(j0_ + 256 * b + r * NR)];
        vv[b][r] = {t.x, t.y};
      }
    }
  }
  if (!regs) __syncthreads();  // all reads of this pass are done before anybody overwrites the buffer
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    cpx (&v)[R] = vv[b];
    const int j = j0_ + 256 * b;
    float2* dst = buf;
  const int k = j & (Ns - 1);
  if (Ns > 1) {
    const int tstep = k * (2048 / (Ns * R));
    const float2 t1 = tw[tstep & 2047], t2 = tw[(2 * tstep) & 2047];
    const cpx w1 = {t1.x, t1.y}, w2 = {t2.x, t2.y};
    const cpx w3 = cmul(w1, w2);
    v[1] = cmul(v[1], w1);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    if (R == 8) {
      const float2 t4 = tw[(4 * tstep) & 2047];
      const cpx w4 = {t4.x, t4.y};
      v[4 % R] = cmul(v[4 % R], w4);
      v[5 % R] = cmul(v[5 % R], cmul(w1, w4));
      v[6 % R] = cmul(v[6 % R], cmul(w2, w4));
      v[7 % R] = cmul(v[7 % R], cmul(w3, w4));
    }
  }
  butterfly(v);
  const int j0 = (j - k) * R + k;
#pragma unroll
  for (int r = 0; r < R; ++r) dst[padi(j0 + r * Ns)] = make_float2(v[r].re, v[r].im);
  }
}

constexpr int kTPair = 1028;    // stride of the two float2 (frame 0, frame 1) weighted-power arrays, >= 1025 bins

__global__ __launch_bounds__(256) void stft_mel_kernel(StftArgs a) {
  __shared__ __attribute__((aligned(16))) float2 buf[kFftLds];  // ONE buffer: 17 kB per workgroup
  __shared__ float wmax[4];
  __shared__ float2 rsum[2][128];  // run sums (frame 0, frame 1): [weight array][run]
  const int tid = threadIdx.x;
  // XCD-aware block -> (clip, frame pair) map (speed only): workgroups are dealt round-robin over the 8 XCDs, so
  // blocks L, L+8, L+16, ... share an L2.  Giving those to consecutive frame pairs of ONE clip lets the 75 %
  // overlap between neighbouring frames hit in that L2 instead of being re-fetched by four different XCDs.
  int u, fp;
  {
    const int npairs = gridDim.x, L = blockIdx.y * gridDim.x + blockIdx.x;
    const int nb = gridDim.y;
    const int full = (nb / 8) * 8 * npairs;  // blocks covered by complete groups of 8 clips
    if (L < full) {
      const int xcd = L & 7, chunk = L >> 3;
      u = (chunk / npairs) * 8 + xcd;
      fp = chunk % npairs;
    } else {
      u = blockIdx.y;
      fp = blockIdx.x;
    }
  }
  const int f0 = fp * 2, f1 = f0 + 1;
  const bool has1 = f1 < a.n_frames;
  const float* yu = a.y + (size_t)u * a.n_y;
  // per-thread constants of the mel stage (L2-resident tables) start their trip now, not after the FFT's last barrier.
  // Wavefront w sums weight array w&1 (lower / upper filter of each bin) over run (w>>1)*64 + lane, both frames at once.
  const int mel_part = (tid >> 6) & 1, mel_run = ((tid >> 7) << 6) + (tid & 63);
  const int mst = a.mel_start[mel_run], mln = a.mel_len[mel_run];
  float mwl[5], mwh[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int k = tid + 256 * i;
    mwl[i] = (k <= 1024) ? a.mel_wlo[k] : 0.0f;
    mwh[i] = (k <= 1024) ? a.mel_whi[k] : 0.0f;
  }
  // frame f covers padded positions [512 f, 512 f + 2048) = y positions [512 f - 1024, ...)
  cpx x0[8];
  if (has1 && f0 >= 2 && f1 * 512 + 1024 <= a.n_y) {
    // both frames lie inside the clip (20 of the 22 pairs of a 1-s clip): no reflection, and frame 1 is frame 0 moved
    // by 512 samples = two of this thread's 256-sample steps, so ten loads feed both (workgroup-uniform branch)
    const float* p = yu + (f0 * 512 - 1024) + tid;
    float sm[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) sm[e] = p[256 * e];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float w = a.hann[tid + 256 * e];
      x0[e] = {w * sm[e], w * sm[e + 2]};
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = tid + 256 * e;
      const float w = a.hann[n];
      const int j0 = f0 * 512 + n - 1024;
      const float s0 = yu[reflect_index(j0, a.n_y)];
      const float s1 = has1 ? yu[reflect_index(j0 + 512, a.n_y)] : 0.0f;
      x0[e] = {w * s0, w * s1};
    }
  }
  if (a.stage_mask & 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) buf[padi(tid + 256 * e)] = make_float2(x0[e].re, x0[e].im);
    __syncthreads();
  } else {
    fft_pass<8, 1>(buf, 1, tid, a.tw, x0);  // straight from registers: no staging write
    __syncthreads();
    fft_pass<8, 1>(buf, 8, tid, a.tw);
    __syncthreads();
    fft_pass<8, 1>(buf, 64, tid, a.tw);
    __syncthreads();
    fft_pass<4, 2>(buf, 512, tid, a.tw);  // radix 4: two butterflies per thread
    __syncthreads();
  }
  // Z = FFT(frame0 + i frame1).  X0[k] = (Z[k] + conj Z[N-k])/2, X1[k] = (Z[k] - conj Z[N-k])/(2i).
  // The powers of bin k (frame 0, frame 1) are multiplied straight away by the bin's two mel weights and stored as
  // PAIRS: Tlo[k] = wlo[k] (P0[k], P1[k]), Thi[k] = whi[k] (P0[k], P1[k]) -- two float2 arrays laid over the same buffer
  // once Z has been read, so that the mel sums below move both frames with one ds_read_b64 + one packed add.
  float2 zz[5], zc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int k = tid + 256 * i;
    if (k <= 1024) {
      zz[i] = buf[padi(k)];
      zc[i] = buf[padi((2048 - k) & 2047)];
    }
  }
  __syncthreads();
  float2* T = buf;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int k = tid + 256 * i;
    if (k <= 1024) {
      const float zr = zz[i].x, zi = zz[i].y, wr = zc[i].x, wi = -zc[i].y;
      const float x0r = 0.5f * (zr + wr), x0i = 0.5f * (zi + wi);
      const float x1r = 0.5f * (zi - wi), x1i = -0.5f * (zr - wr);
      const float p0 = x0r * x0r + x0i * x0i, p1 = x1r * x1r + x1i * x1i;
      const float wl = mwl[i], wh = mwh[i];
      T[k] = make_float2(wl * p0, wl * p1);
      T[kTPair + k] = make_float2(wh * p0, wh * p1);
    }
  }
  __syncthreads();
  // mel[m] = sum over run(m) of Tlo + sum over run(m-1) of Thi.  The kernel is VALU-issue bound, so the sums are
  // arranged for few instructions: one lane per (weight array, run) adds BOTH frames with packed adds, four loads in
  // flight off one address; the (frame, m) threads then pick their two run sums up from a 2 kB exchange array.
  const int sel = tid >> 7, m = tid & 127;
  float s = 0.0f;
  if (!(a.stage_mask & 2)) {
    const float2* Tp = T + mel_part * kTPair + mst;
    float2 a0 = make_float2(0.0f, 0.0f), a1 = a0, a2 = a0, a3 = a0;
    int i = 0;
    for (; i + 4 <= mln; i += 4) {
      const float2 v0 = Tp[i], v1 = Tp[i + 1], v2 = Tp[i + 2], v3 = Tp[i + 3];
      a0.x += v0.x; a0.y += v0.y; a1.x += v1.x; a1.y += v1.y;
      a2.x += v2.x; a2.y += v2.y; a3.x += v3.x; a3.y += v3.y;
    }
    for (; i < mln; ++i) { const float2 v = Tp[i]; a0.x += v.x; a0.y += v.y; }
    rsum[mel_part][mel_run] = make_float2((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y));
    __syncthreads();
    const float* rs = reinterpret_cast<const float*>(&rsum[0][0]);
    s = rs[2 * m + sel] + ((m > 0) ? rs[2 * (128 + m - 1) + sel] : 0.0f);
  } else {
    s = sel ? T[m].y : T[m].x;
  }
  const float dbv = 10.0f * log10f(fmaxf(1e-10f, s));  // librosa.power_to_db(ref=1, amin=1e-10)
  const int f = sel ? f1 : f0;
  if (f < a.n_frames) a.db[((size_t)u * a.n_frames + f) * 128 + m] = dbv;
  const float wm = wave_max(dbv);
  if ((tid & 63) == 0) wmax[tid >> 6] = wm;
  __syncthreads();
  if (tid == 0) a.fmax[(size_t)u * a.n_frames + f0] = fmaxf(wmax[0], wmax[1]);
  if (tid == 128 && has1) a.fmax[(size_t)u * a.n_frames + f1] = fmaxf(wmax[2], wmax[3]);
}

// ---------------------------------------------------------------------------------------------
// stage 2, dual form (round 3): one workgroup = one clip x FOUR frames = two complex FFTs (A = frames f0 + i f1,
// B = f2 + i f3) evaluated by the same threads in lock step, the pair (A, B) in the two halves of every packed-fp32
// operand.  stft_mel_kernel keeps (re, im) of ONE FFT in a packed operand, and half of its vector instructions are the
// half-swaps, negations and moves complex arithmetic needs in that layout (112 v_mov + 57 v_cndmask against 285
// packed math instructions, 686 per wavefront in all).  With (A, B) packed, a complex product is two v_pk_mul + two
// v_pk_fma on plain registers, x(-i) is a register renaming, and one address computation serves both FFTs: about a third
// of the vector instructions per frame.  LDS: one float4 {reA, reB, imA, imB} per point, index e + (e >> 4) (one float4 of
// padding per 16): every access of the four passes is `per-thread base + compile-time offset` -- the XOR swizzle of
// stft_mel_kernel cost three integer instructions per access -- unit-stride ds_read_b128 are conflict-free, and so is the
// stride-8 scatter of pass 1 (8 lanes -> 8 x 4 distinct banks).  34.9 kB + 4 kB per workgroup: four workgroups per CU.
// The two mel weights of a bin multiply the four frames' powers at once (float4 {f0, f2, f1, f3}).
// ---------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void load_tw(const float2* __restrict__ tw, int j, int Ns, cpx (&w)[R - 1]);

typedef float v2f __attribute__((ext_vector_type(2)));
struct cp2 { v2f re, im; };
__device__ __forceinline__ cp2 add2(cp2 a, cp2 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cp2 sub2(cp2 a, cp2 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cp2 mulw(cp2 a, cpx w) { return {a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re}; }
__device__ __forceinline__ cp2 mmi2(cp2 a) { return {a.im, -a.re}; }  // a * (-i)

__device__ __forceinline__ void dft8_2(cp2 (&v)[8]) {
  const float s = 0.70710678118654752440f;
  cp2 a0 = add2(v[0], v[4]), a1 = sub2(v[0], v[4]), a2 = add2(v[2], v[6]), a3 = mmi2(sub2(v[2], v[6]));
  cp2 a4 = add2(v[1], v[5]), a5 = sub2(v[1], v[5]), a6 = add2(v[3], v[7]), a7 = mmi2(sub2(v[3], v[7]));
  cp2 b0 = add2(a0, a2), b2 = sub2(a0, a2), b1 = add2(a1, a3), b3 = sub2(a1, a3);
  cp2 b4 = add2(a4, a6), b6 = sub2(a4, a6), b5 = add2(a5, a7), b7 = sub2(a5, a7);
  cp2 t5 = {(b5.re + b5.im) * s, (b5.im - b5.re) * s};
  cp2 t6 = mmi2(b6);
  cp2 t7 = {(b7.im - b7.re) * s, (-b7.re - b7.im) * s};
  v[0] = add2(b0, b4); v[4] = sub2(b0, b4);
  v[1] = add2(b1, t5); v[5] = sub2(b1, t5);
  v[2] = add2(b2, t6); v[6] = sub2(b2, t6);
  v[3] = add2(b3, t7); v[7] = sub2(b3, t7);
}
__device__ __forceinline__ void dft4_2(cp2 (&v)[4]) {
  cp2 a0 = add2(v[0], v[2]), a1 = sub2(v[0], v[2]), a2 = add2(v[1], v[3]), a3 = mmi2(sub2(v[1], v[3]));
  v[0] = add2(a0, a2); v[2] = sub2(a0, a2); v[1] = add2(a1, a3); v[3] = sub2(a1, a3);
}

constexpr int kF2Buf = 2 * 1028 + 8;  // float4 elements: 2048 points, or the two weighted-power arrays of 1025 bins
constexpr int kF2TP = 1028;                    // stride of the two weighted-power arrays laid over the buffer
__device__ __forceinline__ float4 ld4(const float4* p) { return *p; }
__device__ __forceinline__ void st4(float4* p, cp2 v) { *p = make_float4(v.re.x, v.re.y, v.im.x, v.im.y); }
__device__ __forceinline__ cp2 tocp2(float4 t) { return {v2f{t.x, t.y}, v2f{t.z, t.w}}; }

// Radix-8 Stockham pass of the dual FFT.  rbase / wbase: this thread's padded float4 index of element 0 of its reads
// and writes; the other seven are compile-time offsets (RO(r), WO(r)).
#define LP_F2_PASS8(RO, WO, TW, FIRST)                                                    \
  {                                                                                        \
    cp2 v[8];                                                                              \
    if (FIRST) {                                                                           \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) v[r] = x0[r];                          \
    } else {                                                                               \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) v[r] = tocp2(ld4(rd + (RO(r))));       \
      lds_barrier2();                                                                      \
      _Pragma("unroll") for (int r = 1; r < 8; ++r) v[r] = mulw(v[r], TW[r - 1]);          \
    }                                                                                      \
    dft8_2(v);                                                                             \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) st4(wr + (WO(r)), v[r]);                 \
  }

// Four frames (f0 .. f0 + 3) of clip u: two complex FFTs in packed lock step, powers, mel, dB.  On return thread (pr = tid >> 7,
// m = tid & 127) holds the dB values of mel bin m for frames f0 + 2 pr (dbe) and f0 + 2 pr + 1 (dbo), which it has also
// stored to a.db.  buf / rsum: the workgroup's LDS; every barrier inside is an LDS-only barrier.
__device__ __forceinline__ void stft2_quad(const StftArgs& a, float4* __restrict__ buf, float4 (*__restrict__ rsum)[128],
                                           const float* __restrict__ yu, int u, int f0, int tid, int n_vy, int n_y, int n_frames, float& dbe, float& dbo) {
  const int lane = tid & 63;
  // per-thread constants (L2-resident tables), on their way before the sample loads
  const int mel_part = (tid >> 6) & 1, mel_run = ((tid >> 7) << 6) + lane;
  const int mst = a.mel_start[mel_run], mln = a.mel_len[mel_run];
  float hw[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) hw[e] = a.hann[tid + 256 * e];
  // twiddles of passes 2 and 3 leave now (their L2 round trip hides behind the sample loads and pass 1); those of pass 4 and
  // the mel weights are requested one pass ahead of their use.  The barriers below wait for LDS traffic only.
  cpx w2[7], w3[7];
  load_tw<8>(a.tw, tid, 8, w2);
  load_tw<8>(a.tw, tid, 64, w3);
  cp2 x0[8];
  if (f0 >= 2 && f0 + 3 < n_frames && (f0 + 3) * 512 + 1024 <= n_vy) {
    // all four frames inside the clip: frame j is frame 0 moved by 2 j of the thread's 256-sample steps
    const float* p = yu + (f0 * 512 - 1024) + tid;
    float sm[14];
#pragma unroll
    for (int e = 0; e < 14; ++e) sm[e] = p[256 * e];
#pragma unroll
    for (int e = 0; e < 8; ++e) x0[e] = {v2f{hw[e] * sm[e], hw[e] * sm[e + 4]}, v2f{hw[e] * sm[e + 2], hw[e] * sm[e + 6]}};
  } else {
    // edge quads (2 of a 1-s clip's 11): the generic np.pad index costs a division per sample -- a quarter of the kernel's
    // average instruction count when every edge quad paid it; clips longer than the padding reflect once
    if (n_y > kNFft) {  // workgroup-uniform
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j0 = f0 * 512 + tid + 256 * e - 1024;
        float sj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = reflect_once(j0 + 512 * j, n_y);
            sj[j] = (f0 + j < n_frames && k < n_vy) ? yu[k] : 0.0f;  // [n_vy, n_y): fix_length's zeros
          }
        x0[e] = {v2f{hw[e] * sj[0], hw[e] * sj[2]}, v2f{hw[e] * sj[1], hw[e] * sj[3]}};
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j0 = f0 * 512 + tid + 256 * e - 1024;
        float sj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = reflect_index(j0 + 512 * j, n_y);
            sj[j] = (f0 + j < n_frames && k < n_vy) ? yu[k] : 0.0f;
          }
        x0[e] = {v2f{hw[e] * sj[0], hw[e] * sj[2]}, v2f{hw[e] * sj[1], hw[e] * sj[3]}};
      }
    }
  }
  // ---- four passes.  Element e lives at float4 index swz(e) = e ^ (((e >> 4) & 3) << 1): inside every aligned block
  // of 16 float4 (one 256-byte bank row) a permutation, so unit-stride ds_read_b128 stay conflict-free in the hardware's
  // lane groups, and the stride-8 scatter of pass 1 spreads its 8-lane store groups over 8 different bank quads
  // (scratch/lds_model.py; additive padding made every read 2-way: 29 % of the LDS cycles of the first version).
  // Bits 4-5 of e are bits 4-5 of the thread index for every read and for the writes of passes 3 and 4, so those
  // addresses are `per-thread base + compile-time offset`; passes 1 and 2 pay one v_xor per store.
  const int sx = ((tid >> 4) & 3) << 1;
  const int tsw = tid ^ sx;  // swz(tid + 256 r) = tsw + 256 r
  {
    // pass 1 (Ns = 1): butterfly j = tid writes e = 8 j + r: bits 4-5 of e = bits 1-2 of j
    float4* wr = buf + 8 * tid;
    const int s1 = ((tid >> 1) & 3) << 1;
    const float4* rd = buf;  // (unused: pass 1 takes its inputs from registers)
#define LP_RO1(r) 0
#define LP_WO1(r) ((r) ^ s1)
    const cpx* none = nullptr;
    LP_F2_PASS8(LP_RO1, LP_WO1, none, true)
#undef LP_RO1
#undef LP_WO1
  }
  lds_barrier2();
  {
    // pass 2 (Ns = 8): reads tid + 256 r; writes e = 64 (j >> 3) + k + 8 r: bits 4-5 of e = r >> 1, so the xor value
    // 2 (r >> 1) is a compile-time constant applied to k = j & 7
    const float4* rd = buf + tsw;
    const int k = tid & 7;
    float4* wr = buf + 64 * (tid >> 3);
#define LP_RO2(r) (256 * (r))
#define LP_WO2(r) (8 * (r) + (k ^ (((r) >> 1) << 1)))
    LP_F2_PASS8(LP_RO2, LP_WO2, w2, false)
#undef LP_WO2
  }
  cpx wa[3], wb[3];
  load_tw<4>(a.tw, tid, 512, wa);
  load_tw<4>(a.tw, tid + 256, 512, wb);
  lds_barrier2();
  {
    // pass 3 (Ns = 64): writes e = 512 (j >> 6) + k + 64 r, k = j & 63: bits 4-5 of e = bits 4-5 of k
    const float4* rd = buf + tsw;
    const int k = tid & 63;
    float4* wr = buf + 512 * (tid >> 6) + (k ^ (((k >> 4) & 3) << 1));
#define LP_WO3(r) (64 * (r))
    LP_F2_PASS8(LP_RO2, LP_WO3, w3, false)
#undef LP_WO3
#undef LP_RO2
  }
  float mwl[5], mwh[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int kb = tid + 256 * i;
    mwl[i] = (kb <= 1024) ? a.mel_wlo[kb] : 0.0f;
    mwh[i] = (kb <= 1024) ? a.mel_whi[kb] : 0.0f;
  }
  lds_barrier2();
  // pass 4 (Ns = 512, radix 4): butterflies j = tid and tid + 256 read and write e = j + 512 r.  Its outputs are the
  // spectrum in natural order: bin tid + 256 i of this thread is va[i / 2] (i even) or vb[i / 2] (i odd)
  cp2 va[4], vb[4];
  {
    float4* pa = buf + tsw;
    float4* pb = pa + 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) { va[r] = tocp2(ld4(pa + 512 * r)); vb[r] = tocp2(ld4(pb + 512 * r)); }
    lds_barrier2();
#pragma unroll
    for (int r = 1; r < 4; ++r) { va[r] = mulw(va[r], wa[r - 1]); vb[r] = mulw(vb[r], wb[r - 1]); }
    dft4_2(va);
    dft4_2(vb);
    // only the upper half of the spectrum goes back to LDS: the partners Z[2048 - k] of the bins k <= 1024 live there
#pragma unroll
    for (int r = 2; r < 4; ++r) { st4(pa + 512 * r, va[r]); st4(pb + 512 * r, vb[r]); }
  }
  lds_barrier2();
  // ---- separation by conjugate symmetry, powers, the two mel weights of each bin.  Z[k] is in registers (above); the
  // partner Z[2048 - k] belongs to another thread and comes from LDS (k = 0 is its own partner).
  float4 zc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int kb = tid + 256 * i;
    if (kb <= 1024) {
      const int kc = 2048 - kb;  // 1024 .. 2048
      if (kb == 0) zc[i] = make_float4(va[0].re.x, va[0].re.y, va[0].im.x, va[0].im.y);
      else zc[i] = buf[kc ^ (((kc >> 4) & 3) << 1)];
    }
  }
  lds_barrier2();
  float4* T = buf;  // Tlo[k] at k, Thi[k] at kF2TP + k: {f0, f2, f1, f3} x weight
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int kb = tid + 256 * i;
    if (kb <= 1024) {
      const cp2 z = (i & 1) ? vb[i >> 1] : va[i >> 1];
      const v2f zr = z.re, zi = z.im, cr = {zc[i].x, zc[i].y}, ci = {-zc[i].z, -zc[i].w};
      const v2f x0r = 0.5f * (zr + cr), x0i = 0.5f * (zi + ci), x1r = 0.5f * (zi - ci), x1i = -0.5f * (zr - cr);
      const v2f p0 = x0r * x0r + x0i * x0i, p1 = x1r * x1r + x1i * x1i;  // {f0, f2}, {f1, f3}
      T[kb] = make_float4(mwl[i] * p0.x, mwl[i] * p0.y, mwl[i] * p1.x, mwl[i] * p1.y);
      T[kF2TP + kb] = make_float4(mwh[i] * p0.x, mwh[i] * p0.y, mwh[i] * p1.x, mwh[i] * p1.y);
    }
  }
  lds_barrier2();
  // ---- mel run sums: one lane per (weight array, run), all four frames per load
  {
    const float4* Tp = T + mel_part * kF2TP + mst;
    v2f s0 = {0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int i = 0;
    for (; i + 2 <= mln; i += 2) {
      const float4 t0 = Tp[i], t1 = Tp[i + 1];
      s0 += v2f{t0.x, t0.y}; s1 += v2f{t0.z, t0.w};
      s2 += v2f{t1.x, t1.y}; s3 += v2f{t1.z, t1.w};
    }
    if (i < mln) { const float4 t0 = Tp[i]; s0 += v2f{t0.x, t0.y}; s1 += v2f{t0.z, t0.w}; }
    s0 += s2; s1 += s3;
    rsum[mel_part][mel_run] = make_float4(s0.x, s0.y, s1.x, s1.y);
  }
  lds_barrier2();
  // ---- mel = run(m) of Tlo + run(m - 1) of Thi; thread (pair, m) finishes frames f0 + 2 pair and f0 + 2 pair + 1
  const int pr = tid >> 7, m = tid & 127;
  float4 sum = rsum[0][m];
  if (m > 0) { const float4 h2 = rsum[1][m - 1]; sum.x += h2.x; sum.y += h2.y; sum.z += h2.z; sum.w += h2.w; }
  const float se = pr ? sum.y : sum.x, so = pr ? sum.w : sum.z;  // even / odd frame of the pair
  dbe = 10.0f * log10f(fmaxf(1e-10f, se));
  dbo = 10.0f * log10f(fmaxf(1e-10f, so));
  const int fe = f0 + 2 * pr, fo = fe + 1;
  if (fe < n_frames) a.db[((size_t)u * a.n_frames + fe) * 128 + m] = dbe;
  if (fo < n_frames) a.db[((size_t)u * a.n_frames + fo) * 128 + m] = dbo;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void stft_mel2_kernel(StftArgs a) {
  __shared__ __attribute__((aligned(16))) float4 buf[kF2Buf];
  __shared__ __attribute__((aligned(16))) float4 rsum[2][128];  // run sums {f0, f2, f1, f3}: [weight array][run]
  __shared__ float wmax[4][2];
  const int tid = threadIdx.x, lane = tid & 63;
  // XCD-aware block -> (clip, frame quad) map, as in stft_mel_kernel
  int u, fq;
  {
    const int nq = gridDim.x, L = blockIdx.y * gridDim.x + blockIdx.x, nb = gridDim.y;
    const int full = (nb / 8) * 8 * nq;
    if (L < full) {
      const int chunk = L >> 3;
      u = (chunk / nq) * 8 + (L & 7);
      fq = chunk % nq;
    } else {
      u = blockIdx.y;
      fq = blockIdx.x;
    }
  }
  const int f0 = 4 * fq;
  const float* yu = a.y + (size_t)u * a.n_y;
  int n_y = a.n_y, n_frames = a.n_frames, n_vy = a.n_y;  // (one length for all: y already ends in its zeros)
  if (a.n_valid) {  // this clip's own length: frame count and reflect padding follow it
    clip_lengths(min(max(a.n_valid[u], 0), a.n_samp_max), a.sr_in, &n_vy, &n_y, &n_frames);
    n_frames = min(n_frames, a.n_frames);
  }
  if (f0 >= n_frames) return;  // (workgroup-uniform, before any barrier)
  float dbe, dbo;
  stft2_quad(a, buf, rsum, yu, u, f0, tid, n_vy, n_y, n_frames, dbe, dbo);
  const float me = wave_max(dbe), mo = wave_max(dbo);
  if (lane == 0) { wmax[tid >> 6][0] = me; wmax[tid >> 6][1] = mo; }
  lds_barrier2();
  if (tid < 4) {
    const int f = f0 + tid, w0 = 2 * (tid >> 1), c = tid & 1;
    if (f < n_frames) a.fmax[(size_t)u * a.n_frames + f] = fmaxf(wmax[w0][c], wmax[w0 + 1][c]);
  }
}
#undef LP_F2_PASS8

// ---------------------------------------------------------------------------------------------
// stage 2 for a short window of any length (Speaker recognition/extract_features_construct_dataset.py:224-226:
// librosa.feature.mfcc(win_length=441, n_fft=441, hop_length=220), 1 + 22050/220 = 101 frames): the windowed
// real DFT evaluated as an fp32 MFMA contraction  frames[rows][n_fft] x table[n_fft][re | im].
//
// Layout: every clip is thought of as reflect-padded into rpc*hop floats (rpc = rows per clip), so that global
// frame row r starts at position r*hop for ALL clips: the overlapping frames are just a matrix with leading
// dimension hop; rows frame >= n_frames of a clip are computed and dropped.  The padded layout is virtual -- the
// reflection is applied while a workgroup stages its rows.  One workgroup takes 64 consecutive rows: their
// samples (63*hop + K floats) are staged in LDS once, wavefront t owns the 32 bins of tile t and streams the
// table's 64 columns (re, im) for those bins from L2 in double-buffered groups of 8 K-steps, four
// 32x32x2 MFMA accumulators (2 row blocks x re/im).  Power goes back to LDS, then mel (CSR bank, sequential
// fp32 like the FFT path), dB and the per-frame maximum, one wavefront per row.
// ---------------------------------------------------------------------------------------------
struct DftArgs {
  const float* y;      // [batch][n_y] (unpadded: the reflect padding is applied while the rows are staged)
  const float* table;  // [k_rows][n_tiles*64]
  int n_y, batch;
  int hop, n_fft, k_rows, n_tiles, rpc, n_frames, total_rows;
  const int* mel_start;
  const int* mel_len;
  const int* mel_off;
  const float* mel_w;
  float* db;    // [B][n_frames][128]
  float* fmax;  // [B][n_frames]
};
constexpr int kDftRows = 64, kDftGroup = 8, kDftMaxTiles = 8;

__global__ __launch_bounds__(64 * kDftMaxTiles) __attribute__((amdgpu_waves_per_eu(4, 4))) void dft_mel_kernel(DftArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, n_waves = nthreads >> 6;
  const int row0 = blockIdx.x * kDftRows;
  {
    // position g of the virtual padded layout (clip c at c * rpc * hop, np.pad(y, N/2, 'reflect') inside): 64 rows
    // span at most two clips because rpc > 64 is not required -- the clip index is found per element
    const int n_a = (kDftRows - 1) * a.hop + a.n_fft + 1;
    const int stride = a.rpc * a.hop, pad = a.n_fft / 2;
    const long g0 = (long)row0 * a.hop;
    for (int i = tid; i < n_a; i += nthreads) {
      const long g = g0 + i;
      const int c = (int)(g / stride);
      const int j = (int)(g - (long)c * stride);
      float v = 0.0f;
      if (c < a.batch && j < a.n_y + 2 * pad) {
        int k = j - pad;  // the edge sample is not repeated
        if (k < 0) k = -k;
        else if (k >= a.n_y) k = 2 * (a.n_y - 1) - k;
        v = a.y[(size_t)c * a.n_y + k];
      }
      dsm[i] = v;
    }
  }
  __syncthreads();
  const int li = lane & 31, kk = lane >> 5;
  const int ld = a.n_tiles * 64;
  rs_f32x16 re0, im0, re1, im1;
#pragma unroll
  for (int q = 0; q < 16; ++q) { re0[q] = 0.f; im0[q] = 0.f; re1[q] = 0.f; im1[q] = 0.f; }
  const float* bp = a.table + (size_t)kk * ld + wave * 64 + li;
  // row n of the folded table meets x[n] + x[(N - n) mod N] (real part) and x[n] - x[(N - n) mod N] (imaginary part)
  // (n = 2 s + kk walks forward from x[kk], its partner N - n backward from x[N - kk]; n = 0 meets x[N], one
  // past the frame, under a zero weight (w[0] = 0); padded rows n > N/2 stay inside the frame and meet zero rows)
  const float* fw0 = dsm + li * a.hop + kk;
  const float* bw0 = dsm + li * a.hop + a.n_fft - kk;
  const float* fw1 = fw0 + 32 * a.hop;
  const float* bw1 = bw0 + 32 * a.hop;
  const int n_groups = a.k_rows / (2 * kDftGroup);
  float br0[kDftGroup], bi0[kDftGroup], br1[kDftGroup], bi1[kDftGroup];
#define LP_DFT_LOAD(BR, BI, G)                                          \
  _Pragma("unroll") for (int u = 0; u < kDftGroup; ++u) {               \
    const float* q_ = bp + (size_t)(2 * ((G) * kDftGroup + u)) * ld;    \
    BR[u] = q_[0];                                                      \
    BI[u] = q_[32];                                                     \
  }
#define LP_DFT_MAC(BR, BI, G)                                           \
  _Pragma("unroll") for (int u = 0; u < kDftGroup; ++u) {               \
    const int s_ = 2 * ((G) * kDftGroup + u);                           \
    const float p0_ = fw0[s_], q0_ = bw0[-s_], p1_ = fw1[s_], q1_ = bw1[-s_]; \
    re0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0_ + q0_, BR[u], re0, 0, 0, 0); \
    im0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0_ - q0_, BI[u], im0, 0, 0, 0); \
    re1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p1_ + q1_, BR[u], re1, 0, 0, 0); \
    im1 = __builtin_amdgcn_mfma_f32_32x32x2f32(p1_ - q1_, BI[u], im1, 0, 0, 0); \
  }
  LP_DFT_LOAD(br0, bi0, 0)
  for (int g = 0; g < n_groups; g += 2) {
    if (g + 1 < n_groups) { LP_DFT_LOAD(br1, bi1, g + 1) }
    LP_DFT_MAC(br0, bi0, g)
    if (g + 1 < n_groups) {
      if (g + 2 < n_groups) { LP_DFT_LOAD(br0, bi0, g + 2) }
      LP_DFT_MAC(br1, bi1, g + 1)
    }
  }
#undef LP_DFT_LOAD
#undef LP_DFT_MAC
  __syncthreads();  // every wavefront is done with the staged samples: the buffer becomes the power tile
  const int ldp = a.n_tiles * 32 + 1;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = (q & 3) + 8 * (q >> 2) + 4 * kk;
    dsm[r * ldp + wave * 32 + li] = re0[q] * re0[q] + im0[q] * im0[q];
    dsm[(32 + r) * ldp + wave * 32 + li] = re1[q] * re1[q] + im1[q] * im1[q];
  }
  __syncthreads();
  // mel + dB, one wavefront per frame row: lane handles filters lane and lane + 64
  for (int r = wave; r < kDftRows; r += n_waves) {
    const int grow = row0 + r;
    const int clip = grow / a.rpc, frame = grow - clip * a.rpc;
    if (grow >= a.total_rows || frame >= a.n_frames) continue;  // wave-uniform
    const float* pr = dsm + r * ldp;
    float dbv[2];
#pragma unroll
    for (int hmel = 0; hmel < 2; ++hmel) {
      const int m = lane + 64 * hmel;
      const int st = a.mel_start[m], ln = a.mel_len[m];
      const float* w = a.mel_w + a.mel_off[m];
      float sacc = 0.0f;
      for (int j = 0; j < ln; ++j) sacc = fmaf(w[j], pr[st + j], sacc);
      dbv[hmel] = 10.0f * log10f(fmaxf(1e-10f, sacc));  // librosa.power_to_db(ref=1, amin=1e-10)
    }
    float* dst = a.db + ((size_t)clip * a.n_frames + frame) * 128;
    dst[lane] = dbv[0];
    dst[lane + 64] = dbv[1];
    const float mx = wave_max(fmaxf(dbv[0], dbv[1]));
    if (lane == 0) a.fmax[(size_t)clip * a.n_frames + frame] = mx;
  }
}

// ---------------------------------------------------------------------------------------------
// stages 1 + 2 fused: the resampled signal never leaves the CU.
//
//   One workgroup = (clip, frame group).  A frame group is a run of STFT frames whose reflect-padded windows lie
//   inside 16 consecutive q-blocks of the resampled clip (16 x 441 = 7056 samples; a 1-s clip has the four groups
//   12 | 10 | 10 | 12 frames, built on the host by build_groups()).  The workgroup
//     1. stages the input samples of its 16 q-blocks in LDS (float4 / short4 loads; int16 PCM is scaled by 2^-15 here,
//        which is librosa.load's decode; samples outside [0, n_valid) are zero = resampy's tap-count clamps),
//     2. resamples them with the polyphase contraction of resample_mfma_kernel on v_mfma_f32_16x16x4_f32 -- rows are
//        the 16 q-blocks of ONE clip (A operand from LDS: row stride down + 2 floats, so the 16 rows x 2 k-columns of a
//        half-wave hit 32 different banks), columns 2 x 16 phases sharing one A read, K = the 152-sample band -- and
//        writes the 7056 resampled samples to a second LDS region (fix_length zeros past int(n * ratio)),
//     3. runs the frame pairs of its group as complex 2048-point FFTs straight from that region (the code of
//        stft_mel_kernel; the x staging area becomes the FFT buffer), with the Hann weights, all radix-8 / radix-4
//        twiddles and the mel-stage constants of each thread loaded ONCE per workgroup instead of once per frame pair.
//   HBM traffic of the stage drops from 4.5x to about 1.4x the algorithmic bytes (the resampled signal's 88 kB per clip
//   written and 89 kB read back are gone; the 16-q windows of neighbouring groups overlap by ~30 %, served from L2).
//   The MFMA pipe (resampling) and the VALU (FFT butterflies) belong to different phases of a workgroup; with three
//   workgroups per CU in different phases the two pipes overlap.
//   Clips of different lengths in one launch: n_valid[u] samples of clip u are real, the rest of its row is ignored;
//   lengths, frame count and the reflect padding follow the clip's own length (the group table is the one of the
//   longest clip: frames a shorter clip does not have are skipped).
// ---------------------------------------------------------------------------------------------
constexpr int kFuQ = 16;                                    // q-blocks per workgroup = rows of the 16x16x4 MFMA
constexpr int kFuUp = 441;
constexpr int kFuYLds = kFuQ * kFuUp + 8;                   // resampled span (floats)
constexpr int kFuXMax = kFuQ * 320 + 160;                   // staged input samples at down = 320 (last row's band end)
constexpr int kFuXLds = kFuXMax + 2 * (kFuXMax / 160) + 6;  // + 2 pad floats per q-block (down >= 160)
constexpr int kFuULds = (kFuXLds > 2 * kFftLds ? kFuXLds : 2 * kFftLds);  // union: x staging | FFT buffer
constexpr int kFuLdsFloats = kFuYLds + kFuULds + 2 * 2 * 128 + 8;
typedef float fu_f32x4 __attribute__((ext_vector_type(4)));

struct FusedArgs {
  const void* wav;     // [batch][row_stride] float32 or int16
  long row_stride;     // samples between clips
  int n_samp_max;      // samples per row that may be valid
  const int* n_valid;  // [batch] or null (= n_samp_max everywhere)
  int sr_in, down;
  int vec;             // rows are 16-byte (f32) / 8-byte (i16) aligned: vector loads
  const float* Hband;  // [n_ptiles][kRsBand][32]
  const int* lo;       // [n_ptiles]
  int n_ptiles;
  const int4* groups;
  int n_groups;
  StftArgs st;         // n_y / n_frames of the LONGEST clip: strides of db / fmax
};

// one Stockham pass with the butterfly's twiddles already in registers (w[r-1] = w^r)
template <int R>
__device__ __forceinline__ void fft_pass_regs(float2* __restrict__ buf, int Ns, int j, const cpx (&w)[R - 1], const cpx* __restrict__ regs) {
  constexpr int NR = 2048 / R;
  cpx v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (regs) {
      v[r] = regs[r];
    } else {
      const float2 t = buf[padi(j + r * NR)];
      v[r] = {t.x, t.y};
    }
  }
  if (!regs) {  // LDS-only barrier (see lds_barrier below)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
  }
  if (Ns > 1) {
#pragma unroll
    for (int r = 1; r < R; ++r) v[r] = cmul(v[r], w[r - 1]);
  }
  butterfly(v);
  const int k = j & (Ns - 1);
  const int j0 = (j - k) * R + k;
#pragma unroll
  for (int r = 0; r < R; ++r) buf[padi(j0 + r * Ns)] = make_float2(v[r].re, v[r].im);
}

template <int R>
__device__ __forceinline__ void load_tw(const float2* __restrict__ tw, int j, int Ns, cpx (&w)[R - 1]) {
  const int tstep = (j & (Ns - 1)) * (2048 / (Ns * R));
#pragma unroll
  for (int r = 1; r < R; ++r) {
    const float2 t = tw[(r * tstep) & 2047];
    w[r - 1] = {t.x, t.y};
  }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global access of the
// wave (vmcnt(0)) -- inside the frame loop that would expose the dB stores of the previous pair (a round trip to L2) at
// the next pair's first barrier, once per pair.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
}

template <bool I16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void mfcc_fused_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float fl[];
  float* ys = fl;                     // resampled span
  float* xs = fl + kFuYLds;           // input staging, later the FFT buffer
  float2* buf = reinterpret_cast<float2*>(xs);
  float2* rsum = reinterpret_cast<float2*>(fl + kFuYLds + kFuULds);  // [2][128]
  float* wmax = fl + kFuYLds + kFuULds + 2 * 2 * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block -> (clip, group) map (speed only): the groups of one clip go to one XCD, whose L2 then serves
  // the ~30 % of input samples neighbouring groups share
  int u, g;
  {
    const int ng = gridDim.x, L = blockIdx.y * gridDim.x + blockIdx.x, nb = gridDim.y;
    const int full = (nb / 8) * 8 * ng;
    if (L < full) {
      const int chunk = L >> 3;
      u = (chunk / ng) * 8 + (L & 7);
      g = chunk % ng;
    } else {
      u = blockIdx.y;
      g = blockIdx.x;
    }
  }
  int n = a.n_samp_max;
  if (a.n_valid) n = min(max(a.n_valid[u], 0), a.n_samp_max);
  int n_vy, n_y, n_frames;
  clip_lengths(n, a.sr_in, &n_vy, &n_y, &n_frames);
  const int4 G = a.groups[g];
  const int q0 = G.x, fb = G.y, fe = min(G.z, n_frames);
  if (fb >= fe) return;  // this clip has no frame in the group (workgroup-uniform)
  const int down = a.down;
  // ---- 1. stage the input: element i of the staging area = sample xbase + i, stored at i + 2 (i / down)
  {
    const int xbase = down * q0 - 64;
    const int x_count = kFuQ * down + 160;
    const size_t row = (size_t)u * a.row_stride;
    for (int v = tid; v < x_count / 4; v += 256) {
      const int i4 = 4 * v, sidx = xbase + i4;
      float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
      if (a.vec && sidx >= 0 && sidx + 3 < n) {
        if (I16) {
          const short4 t = *reinterpret_cast<const short4*>(static_cast<const short*>(a.wav) + row + sidx);
          e0 = (float)t.x * (1.0f / 32768.0f); e1 = (float)t.y * (1.0f / 32768.0f);
          e2 = (float)t.z * (1.0f / 32768.0f); e3 = (float)t.w * (1.0f / 32768.0f);
        } else {
          const float4 t = *reinterpret_cast<const float4*>(static_cast<const float*>(a.wav) + row + sidx);
          e0 = t.x; e1 = t.y; e2 = t.z; e3 = t.w;
        }
      } else if (sidx + 3 >= 0 && sidx < n) {
        float e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int sc = sidx + c;
          e[c] = 0.f;
          if (sc >= 0 && sc < n)
            e[c] = I16 ? (float)static_cast<const short*>(a.wav)[row + sc] * (1.0f / 32768.0f) : static_cast<const float*>(a.wav)[row + sc];
        }
        e0 = e[0]; e1 = e[1]; e2 = e[2]; e3 = e[3];
      }
      float* d = xs + i4 + 2 * (i4 / down);  // 8-byte aligned: two ds_write_b64
      *reinterpret_cast<float2*>(d) = make_float2(e0, e1);
      *reinterpret_cast<float2*>(d + 2) = make_float2(e2, e3);
    }
  }
  __syncthreads();
  // ---- 2. polyphase resampling on the matrix cores: Y[q-block i][phase] = X_i[band] . Hband
  {
    const int ir = lane & 15, kk = lane >> 4;
    const float* xrow = xs + (down + 2) * ir;
    for (int r = wave; r < ((a.st.stage_mask & 512) ? 0 : a.n_ptiles); r += 4) {  // (bit 9: profiling, skips the resampling)
      const int lo_r = a.lo[r];
      const float* hb = a.Hband + (size_t)r * kRsBand * 32 + kk * 32 + ir;
      float b0[kRsBand / 4], b1[kRsBand / 4];
#pragma unroll
      for (int s2 = 0; s2 < kRsBand / 4; ++s2) {
        b0[s2] = hb[s2 * 128];
        b1[s2] = hb[s2 * 128 + 16];
      }
      const int t0 = lo_r + 1 + kk;
      const float* pa = xrow + t0;
      const int cross = down - t0;  // band position t0 + 4 s lies in the next q-block's 320 samples once 4 s >= cross
      fu_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < kRsBand / 4; ++s2) {
        const float av = pa[4 * s2 + ((4 * s2 >= cross) ? 2 : 0)];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0[s2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1[s2], acc1, 0, 0, 0);
      }
      // C layout: column (phase) = lane & 15, row (q-block) = 4 (lane >> 4) + e
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        const int ph = 32 * r + 16 * jh + ir;
        if (ph < kFuUp) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int rw = 4 * kk + e;
            const int t = kFuUp * (q0 + rw) + ph;
            const float val = jh ? acc1[e] : acc0[e];
            ys[kFuUp * rw + ph] = (t < n_vy) ? val : 0.0f;
          }
        }
      }
    }
  }
  __syncthreads();
  if (a.st.stage_mask & 256) return;  // (bit 8: profiling, stops before the frames)
  // ---- 3. the frame pairs of the group, from LDS
  const StftArgs& st = a.st;
  const int ybase = kFuUp * q0;
  // per-thread constants, once per workgroup: Hann weights, twiddles of passes 2-4, mel weights and runs
  float hw[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) hw[e] = st.hann[tid + 256 * e];
  cpx w3[7];
  load_tw<8>(st.tw, tid, 64, w3);
  const int mel_part = (tid >> 6) & 1, mel_run = ((tid >> 7) << 6) + (tid & 63);
  const int mst = st.mel_start[mel_run], mln = st.mel_len[mel_run];
  float mwl[5], mwh[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int k = tid + 256 * i;
    mwl[i] = (k <= 1024) ? st.mel_wlo[k] : 0.0f;
    mwh[i] = (k <= 1024) ? st.mel_whi[k] : 0.0f;
  }
  for (int f0 = fb; f0 < fe; f0 += 2) {
    const int f1 = f0 + 1;
    const bool has1 = f1 < fe;  // fe <= n_frames; a pair never straddles two groups (groups hold whole pairs)
    // The thread index is made opaque inside the loop: every LDS address of the five passes depends on it alone, and
    // hoisted out of the loop as invariants those ~90 addresses would push the per-thread tables into scratch.
    int tq = tid;
    asm volatile("" : "+v"(tq));
    // (the twiddles of passes 2 and 4 are fetched per pair, from the L2-resident table, at the top of the iteration: keeping them
    // resident as well pushed 15 registers per lane into scratch -- 63 MB of scratch traffic per 1024 clips)
    cpx w2[7], w4a[3], w4b[3];
    load_tw<8>(st.tw, tq, 8, w2);
    load_tw<4>(st.tw, tq, 512, w4a);
    load_tw<4>(st.tw, tq + 256, 512, w4b);
    cpx x0[8];
    if (has1 && f0 >= 2 && f1 * 512 + 1024 <= n_y) {
      const float* p = ys + (f0 * 512 - 1024 - ybase) + tq;
      float sm[10];
#pragma unroll
      for (int e = 0; e < 10; ++e) sm[e] = p[256 * e];
#pragma unroll
      for (int e = 0; e < 8; ++e) x0[e] = {hw[e] * sm[e], hw[e] * sm[e + 2]};
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j0 = f0 * 512 + tq + 256 * e - 1024;
        const float s0 = ys[reflect_index(j0, n_y) - ybase];
        const float s1 = has1 ? ys[reflect_index(j0 + 512, n_y) - ybase] : 0.0f;
        x0[e] = {hw[e] * s0, hw[e] * s1};
      }
    }
    {
      const cpx none7[7] = {};
      fft_pass_regs<8>(buf, 1, tq, none7, x0);
      lds_barrier();
      fft_pass_regs<8>(buf, 8, tq, w2, nullptr);
      lds_barrier();
      fft_pass_regs<8>(buf, 64, tq, w3, nullptr);
      lds_barrier();
      // radix 4, two butterflies per thread: both read, the workgroup meets, both write
      cpx va[4], vb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float2 ta = buf[padi(tq + r * 512)], tb = buf[padi(tq + 256 + r * 512)];
        va[r] = {ta.x, ta.y};
        vb[r] = {tb.x, tb.y};
      }
      lds_barrier();
#pragma unroll
      for (int r = 1; r < 4; ++r) { va[r] = cmul(va[r], w4a[r - 1]); vb[r] = cmul(vb[r], w4b[r - 1]); }
      dft4(va);
      dft4(vb);
      // Ns = 512: k = j, output index j + r 512
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        buf[padi(tq + r * 512)] = make_float2(va[r].re, va[r].im);
        buf[padi(tq + 256 + r * 512)] = make_float2(vb[r].re, vb[r].im);
      }
      lds_barrier();
    }
    float2 zz[5], zc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int k = tq + 256 * i;
      if (k <= 1024) {
        zz[i] = buf[padi(k)];
        zc[i] = buf[padi((2048 - k) & 2047)];
      }
    }
    lds_barrier();
    float2* T = buf;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int k = tq + 256 * i;
      if (k <= 1024) {
        const float zr = zz[i].x, zi = zz[i].y, wr = zc[i].x, wi = -zc[i].y;
        const float x0r = 0.5f * (zr + wr), x0i = 0.5f * (zi + wi);
        const float x1r = 0.5f * (zi - wi), x1i = -0.5f * (zr - wr);
        const float p0 = x0r * x0r + x0i * x0i, p1 = x1r * x1r + x1i * x1i;
        T[k] = make_float2(mwl[i] * p0, mwl[i] * p1);
        T[kTPair + k] = make_float2(mwh[i] * p0, mwh[i] * p1);
      }
    }
    lds_barrier();
    const int sel = tq >> 7, m = tq & 127;
    {
      const float2* Tp = T + mel_part * kTPair + mst;
      float2 a0 = make_float2(0.0f, 0.0f), a1 = a0, a2 = a0, a3 = a0;
      int i = 0;
      for (; i + 4 <= mln; i += 4) {
        const float2 v0 = Tp[i], v1 = Tp[i + 1], v2 = Tp[i + 2], v3 = Tp[i + 3];
        a0.x += v0.x; a0.y += v0.y; a1.x += v1.x; a1.y += v1.y;
        a2.x += v2.x; a2.y += v2.y; a3.x += v3.x; a3.y += v3.y;
      }
      for (; i < mln; ++i) { const float2 v = Tp[i]; a0.x += v.x; a0.y += v.y; }
      rsum[mel_part * 128 + mel_run] = make_float2((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y));
    }
    lds_barrier();
    const float* rs = reinterpret_cast<const float*>(rsum);
    const float sacc = rs[2 * m + sel] + ((m > 0) ? rs[2 * (128 + m - 1) + sel] : 0.0f);
    const float dbv = 10.0f * log10f(fmaxf(1e-10f, sacc));  // librosa.power_to_db(ref=1, amin=1e-10)
    const int f = sel ? f1 : f0;
    if (f < fe) st.db[((size_t)u * st.n_frames + f) * 128 + m] = dbv;
    const float wm = wave_max(dbv);
    if (lane == 0) wmax[tq >> 6] = wm;
    lds_barrier();
    if (tq == 0) st.fmax[(size_t)u * st.n_frames + f0] = fmaxf(wmax[0], wmax[1]);
    if (tq == 128 && has1) st.fmax[(size_t)u * st.n_frames + f1] = fmaxf(wmax[2], wmax[3]);
  }
}

// ---------------------------------------------------------------------------------------------
// stage 3: top_db floor, DCT, layout
// ---------------------------------------------------------------------------------------------

constexpr int kDctFrames = 64;  // most frames per workgroup (blockIdx.y = chunk): LDS stays <= 33 kB whatever the clip length

// One workgroup = one clip x `chunk` <= 64 output frames, two wavefronts (32 frames each):
//   out[c][t] = sum_m D[c][m] * max(dB[t][m], clipmax - 80)      c < 20 (padded to 32), m < 128
// as a 32 x 32 x 128 contraction per wavefront on v_mfma_f32_32x32x2_f32 (the same ascending-m fp32 fma chain a
// scalar loop would run): the DCT rows are the A operand (64 registers per lane: the table is stored in fragment order,
// [row][k parity][64], so a lane's share is 16 float4 loads, requested together with the dB tile), the clamped dB tile is
// staged transposed in LDS (row stride chunk + 1, odd: conflict-free writes, unit-stride B-operand reads).  The LDS
// image follows the chunk (44 frames: 23 kB, six workgroups per CU instead of four).
// n_frames = frames per clip the db / frame_max arrays are laid out for; with n_valid (clips of different lengths in
// one launch) clip u has its own, smaller count and frames past it are zero columns, as fix_frames pads them
// (extract_features_construct_dataset.py:33-37).
__global__ __launch_bounds__(128) void dct_kernel(const float* __restrict__ db, const float* __restrict__ frame_max,
                                                   int n_frames, int L, int chunk, const float4* __restrict__ dct_frag,
                                                   const double* __restrict__ aff_mean,
                                                   const double* __restrict__ aff_scale, float* __restrict__ out,
                                                   const int* __restrict__ n_valid, int n_samp_max, int sr_in) {
  extern __shared__ float dbs[];  // [128][chunk + 1] = [m][t]
  __shared__ float red[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, u = blockIdx.x;
  const int li = lane & 31, h = lane >> 5;
  const int t0 = blockIdx.y * chunk;                  // first output frame of this chunk
  const int tl = min(chunk, L - t0);                  // output frames of this chunk (incl. zero padding)
  int nf = n_frames;                                  // frames this clip really has
  if (n_valid) {
    int nvy, ny;
    clip_lengths(min(max(n_valid[u], 0), n_samp_max), sr_in, &nvy, &ny, &nf);
    nf = min(nf, n_frames);
  }
  const int tu = max(0, min(nf - t0, tl));            // of which computed from the spectrogram
  const int tp = chunk + 1;
  // dB tile: up to 64 frames x 128 mels = 64 floats per thread, ALL in flight at once (a plain loop keeps one load in
  // flight per thread, and every load here is a cold-L2 round trip: that was 28 of the first kernel's 30 us)
  const float* src = db + ((size_t)u * n_frames + t0) * 128;
  const int n_live = tu * 128;
  float stage[kDctFrames];
#pragma unroll
  for (int j = 0; j < kDctFrames; ++j) {
    const int i = tid + 128 * j;
    stage[j] = (i < n_live) ? src[i] : 0.0f;
  }
  // A operand: lane (li, h) holds D[li][2 s + h], s < 64 (rows >= 20 are zero in the table)
  float4 av4[16];
  const float4* ap = dct_frag + (li * 2 + h) * 16;
#pragma unroll
  for (int s4 = 0; s4 < 16; ++s4) av4[s4] = ap[s4];
  float mx = -INFINITY;
  for (int t = tid; t < nf; t += 128) mx = fmaxf(mx, frame_max[(size_t)u * n_frames + t]);  // whole clip
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  const float thr = fmaxf(red[0], red[1]) - 80.0f;  // top_db = 80
#pragma unroll
  for (int j = 0; j < kDctFrames; ++j) {
    const int i = tid + 128 * j;  // frame j, mel tid: consecutive lanes, consecutive banks
    if (j < chunk) dbs[tid * tp + j] = (i < n_live) ? fmaxf(stage[j], thr) : 0.0f;
  }
  __syncthreads();
  rs_f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  // (a wavefront whose 32 frames lie past the chunk still runs the chain on zeros of its own: columns >= chunk are never
  // stored, and the reads stay inside the image: clamp the column)
  const float* bp = dbs + h * tp + min(wave * 32 + li, chunk - 1);
#pragma unroll
  for (int s4 = 0; s4 < 16; ++s4) {
    const float a4[4] = {av4[s4].x, av4[s4].y, av4[s4].z, av4[s4].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], bp[2 * (4 * s4 + e) * tp], acc, 0, 0, 0);
  }
  // C layout: row (coefficient) = (q & 3) + 8 (q >> 2) + 4 h, column (frame) = li
  const int t = wave * 32 + li;
  const int n_out = kNMfcc * L;
  if (t < tl) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = (q & 3) + 8 * (q >> 2) + 4 * h;
      if (c < kNMfcc) {
        float v = (t < tu) ? acc[q] : 0.0f;
        const int oo = c * L + t0 + t;
        if (aff_mean) v = (float)(((double)v - aff_mean[oo]) / aff_scale[oo]);
        out[(size_t)u * n_out + oo] = v;
      }
    }
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

// Banded taps [n_tiles][kRsBand][32] and first-phase offsets [n_tiles] for the MFMA resampler; false when the
// ratio does not fit its fixed geometry (128 taps, band <= 152 samples, LDS row of 801 floats).
static bool build_band_tables(const Polyphase& pp, std::vector<float>* hb_out, std::vector<int>* lo_out) {
  if (!(pp.taps == 128 && pp.left == 64 && pp.down + 128 <= kRsStride - 1)) return false;
  const int nt = (pp.up + 31) / 32;
  std::vector<float> hb((size_t)nt * kRsBand * 32, 0.0f);
  std::vector<int> lo(nt, 0);
  for (int r = 0; r < nt; ++r) {
    const int p0 = 32 * r;
    lo[r] = pp.n_off[p0];
    for (int j = 0; j < 32; ++j) {
      const int ph = p0 + j;
      if (ph >= pp.up) break;
      const int d = pp.n_off[ph] - pp.n_off[p0];
      if (d < 0 || d + 128 > kRsBand) return false;
      for (int t = 0; t < 128; ++t) hb[((size_t)r * kRsBand + d + t) * 32 + j] = pp.h[(size_t)ph * 128 + t];
    }
    // the last sample a workgroup's band can touch must stay inside its LDS row
    if (lo[r] + 1 + kRsBand > kRsStride) return false;
  }
  *hb_out = hb;
  *lo_out = lo;
  return true;
}

// Frame groups of a clip with n_y resampled samples and n_frames frames for mfcc_fused_kernel: runs of whole frame
// pairs whose reflect-padded windows (plus one sample of slack below: a SHORTER clip in the same launch reflects around
// its own end and may touch one sample before the window) lie inside kFuQ q-blocks starting at q0.
static std::vector<int> build_groups(int n_y, int n_frames, int up) {
  std::vector<int> g;
  int f = 0;
  while (f < n_frames) {
    const int lo = std::max(0, kHop * f - kNFft / 2 - 1);
    const int q0 = lo / up;
    int fe = f;
    while (fe < n_frames) {
      const int cand = std::min(fe + 2, n_frames);
      const int hi = std::min(n_y, kHop * (cand - 1) + kNFft / 2);  // one past the last sample the frames need
      if ((hi + up - 1) / up - q0 > kFuQ) break;
      fe = cand;
    }
    if (fe == f) return std::vector<int>();  // a single pair does not fit: never with 2048/512 and up = 441
    g.push_back(q0); g.push_back(f); g.push_back(fe); g.push_back(0);
    f = fe;
  }
  return g;
}

// fp16 hi / lo fragments of the banded taps in the B-operand lane order of v_mfma_f32_32x32x16_f16: lane (col, h) of k-step c
// holds B[k = 16 c + 8 h + j][col], j < 8; the band starts at the window position (lo + 1) & ~7
static std::vector<unsigned int> build_band_h2(const std::vector<float>& hb, const std::vector<int>& lo) {
  const int nt = (int)lo.size();
  std::vector<unsigned int> out((size_t)nt * 2 * kRhChunks * 64 * 4, 0u);
  for (int r = 0; r < nt; ++r) {
    const int first = lo[r] + 1, band0 = first & ~7;
    for (int c = 0; c < kRhChunks; ++c)
      for (int ln = 0; ln < 64; ++ln) {
        const int col = ln & 31, hh = ln >> 5;
        unsigned short hi[8], lw[8];
        for (int j = 0; j < 8; ++j) {
          const int kk = band0 + 16 * c + 8 * hh + j - first;  // index into the 152-sample band of the fp32 table
          const float v = (kk >= 0 && kk < kRsBand) ? hb[((size_t)r * kRsBand + kk) * 32 + col] * kRhTapScale : 0.0f;
          const _Float16 a = (_Float16)v;
          const _Float16 b = (_Float16)(v - (float)a);
          memcpy(&hi[j], &a, 2);
          memcpy(&lw[j], &b, 2);
        }
        unsigned int* dh = &out[((((size_t)r * 2 + 0) * kRhChunks + c) * 64 + ln) * 4];
        unsigned int* dl = &out[((((size_t)r * 2 + 1) * kRhChunks + c) * 64 + ln) * 4];
        for (int w = 0; w < 4; ++w) {
          dh[w] = (unsigned int)hi[2 * w] | ((unsigned int)hi[2 * w + 1] << 16);
          dl[w] = (unsigned int)lw[2 * w] | ((unsigned int)lw[2 * w + 1] << 16);
        }
      }
  }
  return out;
}

static int lo_max(const MfccPlan*, const Polyphase& pp) {
  int m = 0;
  for (int r = 0; 32 * r < pp.up; ++r) m = std::max(m, pp.n_off[32 * r]);
  return m;
}

template <typename T>
static int upload(T** dptr, const std::vector<T>& v) {
  LP_HIP(hipMalloc(dptr, v.size() * sizeof(T)));
  LP_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return LIPASR_OK;
}

// the fp16-plane persistent resampler is the one that takes int16 PCM and per-clip lengths
static bool resample_h2_ok(const MfccPlan* p, const void* wav, int fmt) {
  const uintptr_t addr = reinterpret_cast<uintptr_t>(wav);
  const bool vec4 = ((p->n_samp & 3) == 0) && ((p->down & 3) == 0) && ((addr & (fmt ? 7 : 15)) == 0);
  const int n_waves = p->n_ptiles;
  return !p->identity && p->d_hband && p->d_hbandh && !(p->stage_mask & (4 | 16)) && vec4 && n_waves >= 8 && n_waves <= kRpMaxWaves &&
         32 * ((kRsStride - 1) / 4) <= 5 * 64 * n_waves && p->left == 64 && p->down + 128 + 32 <= kRhRowHalfs;
}

static int launch_resample(const MfccPlan* p, const void* wav_any, int fmt, const int* n_valid, int batch, float* y, hipStream_t st) {
  if (resample_h2_ok(p, wav_any, fmt)) {
    const int nq = (p->n_y + p->up - 1) / p->up, tiles = (batch + 31) / 32;
    int wgs = p->rs_target_wgs;  // one workgroup per CU this stream may use
    if (wgs < tiles) wgs = tiles;
    if (wgs > tiles * nq) wgs = tiles * nq;
    const size_t ldsh = (size_t)2 * 32 * kRhRowBytes;
    using kern_t = void (*)(const void*, const int*, int, int, int, float*, int, int, int, int, int, int, int, const uint4*, const int*, int, int);
    static const kern_t kerns[4] = {resample_persist_h2_kernel<false, false>, resample_persist_h2_kernel<true, false>,
                                    resample_persist_h2_kernel<false, true>, resample_persist_h2_kernel<true, true>};
    static bool attr_h = false;
    if (!attr_h) {
      for (kern_t k : kerns)
        LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh));
      attr_h = true;
    }
    hipLaunchKernelGGL(kerns[(fmt ? 1 : 0) + (n_valid ? 2 : 0)], dim3(wgs), dim3(64 * kRpMaxWaves), ldsh, st, wav_any, n_valid,
                       p->sr_in, p->n_samp, batch, y, p->n_valid, p->n_y, p->up, p->down, p->left, nq, tiles,
                       reinterpret_cast<const uint4*>(p->d_hbandh), p->d_lo, p->n_ptiles, (p->stage_mask >> 16) & 7);
    LP_LAUNCH_CHECK();
    return LIPASR_OK;
  }
  if (fmt != 0 || n_valid) {
    set_error("lipasr_mfcc: int16 input and per-clip lengths need the fp16-plane resampler (a rational ratio with <= 16 phase tiles, "
              "row length a multiple of 4); this plan is %d Hz with rows of %d", p->sr_in, p->n_samp);
    return LIPASR_EUNSUPPORTED;
  }
  const float* wav = static_cast<const float*>(wav_any);
  if (p->identity) {
    hipLaunchKernelGGL(copy_pad_kernel, dim3(32, batch), dim3(256), 0, st, wav, p->n_samp, y, p->n_y);
    LP_LAUNCH_CHECK();
    return LIPASR_OK;
  }
  const int nq = (p->n_y + p->up - 1) / p->up;
  const bool vec4 = ((p->n_samp & 3) == 0) && ((p->down & 3) == 0) && ((reinterpret_cast<uintptr_t>(wav) & 15) == 0);
  const int n_waves = p->n_ptiles;
  if (p->d_hband && !(p->stage_mask & 4) && vec4 && n_waves >= 8 && n_waves <= kRpMaxWaves &&
      32 * ((kRsStride - 1) / 4) <= 5 * 64 * n_waves) {
    // persistent fp32 form (the parity reference of the fp16-plane kernel): one workgroup per CU-sized share of the work
    const size_t lds = (size_t)2 * 32 * kRsStride * sizeof(float);
    static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];  /* per device */
    if (!attr_set) {
      LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resample_persist_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    const int tiles = (batch + 31) / 32;
    int wgs = p->rs_target_wgs;
    if (wgs < tiles) wgs = tiles;
    if (wgs > tiles * nq) wgs = tiles * nq;
    hipLaunchKernelGGL(resample_persist_kernel, dim3(wgs), dim3(64 * n_waves), lds, st, wav, p->n_samp, batch, y,
                       p->n_valid, p->n_y, p->up, p->down, p->left, nq, tiles, p->d_hband, p->d_lo);
  } else if (p->d_hband && !(p->stage_mask & 4)) {
    const size_t lds = (size_t)32 * kRsStride * sizeof(float);
    static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];  /* per device */
    if (!attr_set) {
      LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resample_mfma_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    hipLaunchKernelGGL(resample_mfma_kernel, dim3(nq, (batch + 31) / 32), dim3(64 * kRsWaves), lds, st, wav,
                       p->n_samp, batch, y, p->n_valid, p->n_y, p->up, p->down, p->left, p->n_ptiles, p->d_hband, p->d_lo, p->stage_mask);
  } else if (p->taps == 128 && p->up <= 448) {
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

// dct_kernel's A operand: lane (row li, k parity h) reads D[li][2 s + h], s < 64, as 16 float4
static std::vector<float> dct_fragments() {
  const std::vector<float> d = tables::dct_matrix();  // [20][128]
  std::vector<float> f((size_t)32 * 2 * 64, 0.0f);
  for (int li = 0; li < kNMfcc; ++li)
    for (int h = 0; h < 2; ++h)
      for (int s = 0; s < 64; ++s) f[((size_t)li * 2 + h) * 64 + s] = d[(size_t)li * 128 + 2 * s + h];
  return f;
}

static int launch_dct(const MfccPlan* p, int batch, int L, const double* am, const double* as, float* out, const int* n_valid,
                      hipStream_t st) {
  const int chunk = std::min(kDctFrames, (L + 3) & ~3);  // even, so that the LDS row stride chunk + 1 is odd
  const size_t lds = (size_t)128 * (chunk + 1) * sizeof(float);
  hipLaunchKernelGGL(dct_kernel, dim3(batch, (L + chunk - 1) / chunk), dim3(128), lds, st, p->d_db, p->d_fmax, p->n_frames, L, chunk,
                     reinterpret_cast<const float4*>(p->d_dct), am, as, out, n_valid, p->n_samp, p->sr_in);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

static void fill_stft_args(const MfccPlan* p, const float* y, StftArgs* a) {
  a->y = y; a->n_y = p->n_y; a->n_frames = p->n_frames; a->hann = p->d_hann;
  a->tw = reinterpret_cast<const float2*>(p->d_tw);
  a->mel_wlo = p->d_mel_wlo; a->mel_whi = p->d_mel_whi; a->mel_start = p->d_mel_pstart; a->mel_len = p->d_mel_plen;
  a->db = p->d_db; a->fmax = p->d_fmax;
  a->stage_mask = p->stage_mask;
  a->n_valid = nullptr; a->sr_in = p->sr_in; a->n_samp_max = p->n_samp;
}

// stages 1 + 2 in one kernel (mfcc_fused_kernel); wav: float32 (fmt 0) or int16 PCM (fmt 1)
static int launch_fused(const MfccPlan* p, const void* wav, int fmt, const int* n_valid, int batch, hipStream_t st) {
  FusedArgs a;
  a.wav = wav; a.row_stride = p->n_samp; a.n_samp_max = p->n_samp; a.n_valid = n_valid;
  a.sr_in = p->sr_in; a.down = p->down;
  const uintptr_t addr = reinterpret_cast<uintptr_t>(wav);
  a.vec = fmt ? ((addr & 7) == 0 && (p->n_samp & 3) == 0) : ((addr & 15) == 0 && (p->n_samp & 3) == 0);
  a.Hband = p->d_hband; a.lo = p->d_lo; a.n_ptiles = p->n_ptiles;
  a.groups = reinterpret_cast<const int4*>(p->d_groups); a.n_groups = p->n_groups;
  fill_stft_args(p, nullptr, &a.st);
  const size_t lds = (size_t)kFuLdsFloats * sizeof(float);
  static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];  /* per device */
  if (!attr_set) {
    LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfcc_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfcc_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (fmt) hipLaunchKernelGGL(mfcc_fused_kernel<true>, dim3(p->n_groups, batch), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(mfcc_fused_kernel<false>, dim3(p->n_groups, batch), dim3(256), lds, st, a);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

static bool stft2_ok(const MfccPlan* p) { return !p->dft && !(p->stage_mask & (64 | 3)); }

static int launch_from_22k(const MfccPlan* p, const float* y, const int* n_valid, int batch, int L, const double* am, const double* as,
                           float* out, hipStream_t st, hipEvent_t mid = nullptr) {  // mid: recorded between stft_mel and dct
  StftArgs a;
  fill_stft_args(p, y, &a);
  a.n_valid = n_valid; a.sr_in = p->sr_in; a.n_samp_max = p->n_samp;
  if (n_valid && !stft2_ok(p)) {
    set_error("lipasr_mfcc: per-clip lengths need the 2048/512 STFT kernel; this plan has n_fft %d", p->n_fft);
    return LIPASR_EUNSUPPORTED;
  }
  if (p->dft) {
    DftArgs d;
    d.y = y; d.n_y = p->n_y; d.batch = batch; d.table = p->d_dft; d.hop = p->hop; d.n_fft = p->n_fft; d.k_rows = p->dft_krows; d.n_tiles = p->dft_tiles;
    d.rpc = p->dft_rpc; d.n_frames = p->n_frames; d.total_rows = batch * p->dft_rpc;
    d.mel_start = p->d_mel_start; d.mel_len = p->d_mel_len; d.mel_off = p->d_mel_off; d.mel_w = p->d_mel_w;
    d.db = p->d_db; d.fmax = p->d_fmax;
    const int n_a = (kDftRows - 1) * p->hop + p->n_fft + 1, n_p = kDftRows * (p->dft_tiles * 32 + 1);
    const size_t dl = (size_t)(n_a > n_p ? n_a : n_p) * sizeof(float);
    if (dl > 48 * 1024)
      LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dft_mel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)dl));
    hipLaunchKernelGGL(dft_mel_kernel, dim3((d.total_rows + kDftRows - 1) / kDftRows), dim3(64 * p->dft_tiles), dl, st, d);
  } else if (stft2_ok(p) && p->bd.cfrag && !(p->stage_mask & 256)) {
    // one workgroup per clip: on request (plan key 4) the kernel finishes with the top_db floor and the DCT itself
    const bool fuse = p->bd_fuse_dct && bdft_can_fuse_dct(p->n_frames, p->bd_seg, L);
    BdftDct d;
    d.L = L; d.dct_frag = reinterpret_cast<const float4*>(p->d_dct); d.aff_mean = am; d.aff_scale = as; d.out = out;
    const int rc = launch_stft_bdft(a, p->bd, batch, p->bd_seg, fuse ? &d : nullptr, st);
    if (rc != LIPASR_OK) return rc;
    if (fuse) {
      if (mid) LP_HIP(hipEventRecord(mid, st));
      return LIPASR_OK;
    }
  } else if (stft2_ok(p)) {
    hipLaunchKernelGGL(stft_mel2_kernel, dim3((p->n_frames + 3) / 4, batch), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(stft_mel_kernel, dim3((p->n_frames + 1) / 2, batch), dim3(256), 0, st, a);
  }
  LP_LAUNCH_CHECK();
  if (mid) LP_HIP(hipEventRecord(mid, st));
  return launch_dct(p, batch, L, am, as, out, n_valid, st);
}

}  // namespace lipasr

using namespace lipasr;

// opaque plan type of the C ABI
struct lipasr_mfcc : lipasr::MfccPlan {};

namespace lipasr {

static void plan_unregister(MfccPlan* p) {
  if (!p || !p->ctx) return;
  std::vector<MfccPlan*>& v = p->ctx->mfcc_plans;
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == p) { v.erase(v.begin() + i); break; }
  if (p->ctx->mfcc == p) p->ctx->mfcc = nullptr;
}

static int plan_build(lipasr_handle_t h, int sr_in, int n_samp, int batch_max, int n_fft, int hop, MfccPlan** out) {
  LP_CHECK_ARG(h != nullptr && out != nullptr, "lipasr_mfcc_plan: null argument");
  const bool dft = !(n_fft == kNFft && hop == kHop);
  if (dft && !(n_fft >= 32 && n_fft <= 32 * kDftMaxTiles * 2 - 2 && hop >= 1 && hop <= n_fft)) {
    set_error("lipasr_mfcc_plan_ex: n_fft=%d hop=%d; supported: 2048/512 (FFT path) or 32 <= n_fft <= %d with "
              "1 <= hop <= n_fft (DFT-contraction path)", n_fft, hop, 32 * kDftMaxTiles * 2 - 2);
    return LIPASR_EUNSUPPORTED;
  }
  LP_CHECK_ARG(sr_in >= 1000 && sr_in <= 384000, "lipasr_mfcc_plan: sr_in=%d outside [1000, 384000]", sr_in);
  LP_CHECK_ARG(n_samp >= 2 && batch_max >= 1, "lipasr_mfcc_plan: n_samp=%d batch_max=%d", n_samp, batch_max);
  DeviceGuard g(h->device);
  MfccPlan* p = new lipasr_mfcc();
  p->ctx = h;
  p->sr_in = sr_in; p->n_samp = n_samp; p->batch_max = batch_max;
  p->identity = (sr_in == kSr);
  resampled_lengths(n_samp, sr_in, kSr, &p->n_valid, &p->n_y);
  if (p->n_y < 2) { delete p; set_error("lipasr_mfcc_plan: clip too short after resampling"); return LIPASR_EINVAL; }
  p->n_fft = n_fft; p->hop = hop; p->dft = dft;
  p->rs_target_wgs = h->rs_target_wgs;
  p->n_frames = 1 + p->n_y / hop;
  // the FFT path reflects repeatedly like np.pad (any clip of >= 2 samples); the short-window path's pad kernel
  // reflects once, which needs the clip to be longer than the padding
  if (dft && p->n_y <= n_fft / 2) { delete p; set_error("lipasr_mfcc_plan_ex: clip shorter than the reflect padding"); return LIPASR_EINVAL; }
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
    {
      std::vector<float> hb;
      std::vector<int> lo;
      if (build_band_tables(pp, &hb, &lo)) {
        p->n_ptiles = (int)lo.size();
        if ((rc = upload(&p->d_hband, hb)) != LIPASR_OK || (rc = upload(&p->d_lo, lo)) != LIPASR_OK) {
          mfcc_plan_free(p);
          return rc;
        }
        // fp16-plane fragments for resample_persist_h2_kernel: the band of every tile must fit 160 samples from its
        // 8-aligned start and stay inside the 480-sample window
        bool fits = true;
        for (int r = 0; r < (int)lo.size(); ++r) {
          const int first = lo[r] + 1, band0 = first & ~7;
          fits = fits && (first - band0 + kRsBand <= kRhK) && (band0 + kRhK <= kRhRowHalfs);
        }
        if (fits && (rc = upload(&p->d_hbandh, build_band_h2(hb, lo))) != LIPASR_OK) { mfcc_plan_free(p); return rc; }
      }
    }
    // the fused resample -> STFT kernel: 2048/512 frames, 441 phases (16 kHz and 8 kHz input), rows 2 banks apart
    if (!dft && p->d_hband && pp.up == kFuUp && pp.left == 64 && (pp.down % 32) == 0 && pp.down >= 160 && pp.down <= 320 &&
        (kFuQ - 1) * pp.down + lo_max(p, pp) + 1 + kRsBand <= kFuQ * pp.down + 160) {
      std::vector<int> groups = build_groups(p->n_y, p->n_frames, pp.up);
      if (!groups.empty()) {
        p->n_groups = (int)groups.size() / 4;
        if ((rc = upload(&p->d_groups, groups)) != LIPASR_OK) { mfcc_plan_free(p); return rc; }
        p->fused = true;
      }
    }
  }
  MelSparse ms = mel_sparse(n_fft);
  if (dft) {
    p->dft_tiles = (1 + n_fft / 2 + 31) / 32;
    p->dft_krows = ((n_fft / 2 + 1 + 2 * kDftGroup - 1) / (2 * kDftGroup)) * (2 * kDftGroup);  // folded: rows 0..N/2
    p->dft_rpc = (p->n_y + 2 * (n_fft / 2) + hop - 1) / hop;
    if ((rc = upload(&p->d_dft, dft_table(n_fft, p->dft_krows, p->dft_tiles))) != LIPASR_OK) { mfcc_plan_free(p); return rc; }
  }
  MelPairs mp = dft ? MelPairs() : mel_pairs();
  if (dft) { mp.wlo.assign(1, 0.f); mp.whi.assign(1, 0.f); mp.start.assign(1, 0); mp.len.assign(1, 0); }
  if (!mp.ok) {
    mfcc_plan_free(p);
    set_error("lipasr_mfcc_plan: mel filter bank is not a two-filters-per-bin bank");
    return LIPASR_EUNSUPPORTED;
  }
  if ((rc = upload(&p->d_mel_wlo, mp.wlo)) != LIPASR_OK || (rc = upload(&p->d_mel_whi, mp.whi)) != LIPASR_OK ||
      (rc = upload(&p->d_mel_pstart, mp.start)) != LIPASR_OK || (rc = upload(&p->d_mel_plen, mp.len)) != LIPASR_OK) {
    mfcc_plan_free(p);
    return rc;
  }
  if ((rc = upload(&p->d_hann, hann_periodic())) != LIPASR_OK || (rc = upload(&p->d_tw, twiddles())) != LIPASR_OK ||
      (rc = upload(&p->d_mel_start, ms.start)) != LIPASR_OK || (rc = upload(&p->d_mel_len, ms.len)) != LIPASR_OK ||
      (rc = upload(&p->d_mel_off, ms.off)) != LIPASR_OK || (rc = upload(&p->d_mel_w, ms.w)) != LIPASR_OK ||
      (rc = upload(&p->d_dct, dct_fragments())) != LIPASR_OK) {
    mfcc_plan_free(p);
    return rc;
  }
  if (!dft && (rc = bdft_tables_build(&p->bd)) != LIPASR_OK) { mfcc_plan_free(p); return rc; }
  const size_t ny = (size_t)batch_max * p->n_y, ndb = (size_t)batch_max * p->n_frames * 128,
               nfm = (size_t)batch_max * p->n_frames;
  if (hipMalloc(&p->d_y, ny * sizeof(float)) != hipSuccess || hipMalloc(&p->d_db, ndb * sizeof(float)) != hipSuccess ||
      hipMalloc(&p->d_fmax, nfm * sizeof(float)) != hipSuccess) {
    mfcc_plan_free(p);
    set_error("lipasr_mfcc_plan: intermediate allocation failed");
    return LIPASR_ENOMEM;
  }
  h->mfcc_plans.push_back(p);
  *out = p;
  return LIPASR_OK;
}

static int plan_check(const char* fn, const MfccPlan* p, int batch, int L) {
  LP_CHECK_ARG(p != nullptr, "%s: null plan", fn);
  LP_CHECK_ARG(batch >= 1 && batch <= p->batch_max, "%s: batch %d outside [1, %d]", fn, batch, p->batch_max);
  LP_CHECK_ARG(L >= 1, "%s: utterance_length=%d", fn, L);
  return LIPASR_OK;
}

static int plan_resample(MfccPlan* p, const float* wav, int batch, float* y, hipStream_t st) {
  hipEvent_t* ev = (p->prof_n < p->prof_cap) ? &p->prof_events[5 * (size_t)p->prof_n] : nullptr;
  if (ev) LP_HIP(hipEventRecord(ev[0], st));
  int rc = launch_resample(p, wav, 0, nullptr, batch, y, st);
  if (rc != LIPASR_OK) return rc;
  if (ev) {
    LP_HIP(hipEventRecord(ev[1], st));
    p->prof_half = true;
  }
  return LIPASR_OK;
}

static int plan_from_22k(MfccPlan* p, const float* y, int batch, int n_y, int L, const double* am, const double* as, float* out,
                         hipStream_t st) {
  LP_CHECK_ARG(n_y == p->n_y, "lipasr_mfcc_from_22k: n_y=%d but the plan was made for %d", n_y, p->n_y);
  LP_CHECK_ARG((am == nullptr) == (as == nullptr), "lipasr_mfcc_from_22k: give both affine arrays or neither");
  // timed only as the second half of a split extraction (a resample timing is already in the slot)
  hipEvent_t* ev = (p->prof_half && p->prof_n < p->prof_cap) ? &p->prof_events[5 * (size_t)p->prof_n] : nullptr;
  if (ev) LP_HIP(hipEventRecord(ev[2], st));
  int rc = launch_from_22k(p, y, nullptr, batch, L, am, as, out, st, ev ? ev[3] : nullptr);
  if (rc != LIPASR_OK) return rc;
  if (ev) {
    LP_HIP(hipEventRecord(ev[4], st));
    p->prof_half = false;
    p->prof_n++;
  }
  return LIPASR_OK;
}

// the whole extraction.  fmt 0: float32 samples, 1: int16 PCM.  n_valid: per-clip sample counts (device) or null.
static int plan_run(MfccPlan* p, const void* wav, int fmt, const int* n_valid, int batch, int L, const double* am, const double* as,
                    float* out, hipStream_t st) {
  LP_CHECK_ARG(wav && out, "lipasr_mfcc: null argument");
  LP_CHECK_ARG(fmt == 0 || fmt == 1, "lipasr_mfcc: sample format %d (0 = float32, 1 = int16 PCM)", fmt);
  LP_CHECK_ARG((am == nullptr) == (as == nullptr), "lipasr_mfcc: give both affine arrays or neither");
  // three kernels (resample -> y in HBM -> STFT+mel -> DCT) unless the plan prefers the single fused resample+STFT kernel, which
  // moves 2.6x fewer bytes and is slower; the fused kernel also takes over when the three-kernel form cannot read
  // this input (unaligned int16 / ragged rows)
  const bool can_fuse = p->fused && !(p->stage_mask & 128);
  const bool three_ok = (fmt == 0 && !n_valid) || (resample_h2_ok(p, wav, fmt) && stft2_ok(p));
  const bool fused = can_fuse && (p->prefer_fused || !three_ok);
  if (!fused && !three_ok) {
    set_error("lipasr_mfcc: int16 input and per-clip lengths need the 2048/512 path with a 441/320- or 441/160-style resampler "
              "(16 kHz or 8 kHz input, rows a multiple of 4 samples); this plan is %d Hz, n_fft %d, rows of %d", p->sr_in, p->n_fft, p->n_samp);
    return LIPASR_EUNSUPPORTED;
  }
  hipEvent_t* ev = (!p->prof_half && p->prof_n < p->prof_cap) ? &p->prof_events[5 * (size_t)p->prof_n] : nullptr;
  int rc;
  if (fused) {
    if (ev) {  // no separate resampling kernel: its slot stays empty, the fused kernel is timed as stft_mel
      LP_HIP(hipEventRecord(ev[0], st));
      LP_HIP(hipEventRecord(ev[1], st));
      LP_HIP(hipEventRecord(ev[2], st));
    }
    if ((rc = launch_fused(p, wav, fmt, n_valid, batch, st)) != LIPASR_OK) return rc;
    if (ev) LP_HIP(hipEventRecord(ev[3], st));
    if ((rc = launch_dct(p, batch, L, am, as, out, n_valid, st)) != LIPASR_OK) return rc;
  } else {
    if (ev) LP_HIP(hipEventRecord(ev[0], st));
    if ((rc = launch_resample(p, wav, fmt, n_valid, batch, p->d_y, st)) != LIPASR_OK) return rc;
    if (ev) {
      LP_HIP(hipEventRecord(ev[1], st));
      LP_HIP(hipEventRecord(ev[2], st));
    }
    if ((rc = launch_from_22k(p, p->d_y, n_valid, batch, L, am, as, out, st, ev ? ev[3] : nullptr)) != LIPASR_OK) return rc;
  }
  if (ev) {
    LP_HIP(hipEventRecord(ev[4], st));
    p->prof_n++;
  }
  return LIPASR_OK;
}

static int plan_profile_begin(MfccPlan* p, int max_calls) {
  LP_CHECK_ARG(p != nullptr && max_calls >= 1 && max_calls <= 100000, "lipasr_mfcc_profile_begin: bad argument");
  DeviceGuard g(p->ctx->device);
  while ((int)p->prof_events.size() < 5 * max_calls) {
    hipEvent_t e;
    LP_HIP(hipEventCreate(&e));
    p->prof_events.push_back(e);
  }
  p->prof_cap = max_calls;
  p->prof_n = 0;
  p->prof_half = false;
  return LIPASR_OK;
}

static int plan_profile_end(MfccPlan* p, float* avg_ms3, int* n_calls) {
  LP_CHECK_ARG(p && avg_ms3 && n_calls, "lipasr_mfcc_profile_end: null argument");
  double acc[3] = {0, 0, 0};
  static const int kFrom[3] = {0, 2, 3}, kTo[3] = {1, 3, 4};
  for (int i = 0; i < p->prof_n; ++i) {
    hipEvent_t* ev = &p->prof_events[5 * (size_t)i];
    LP_HIP(hipEventSynchronize(ev[4]));
    for (int k = 0; k < 3; ++k) {
      float ms = 0.0f;
      LP_HIP(hipEventElapsedTime(&ms, ev[kFrom[k]], ev[kTo[k]]));
      acc[k] += ms;
    }
  }
  *n_calls = p->prof_n;
  for (int k = 0; k < 3; ++k) avg_ms3[k] = p->prof_n ? (float)(acc[k] / p->prof_n) : 0.0f;
  p->prof_cap = 0;
  p->prof_n = 0;
  p->prof_half = false;
  return LIPASR_OK;
}

static int plan_set(MfccPlan* p, int key, int value) {
  LP_CHECK_ARG(p != nullptr, "lipasr_mfcc_set: null plan");
  LP_CHECK_ARG(key >= 0 && key <= 4, "lipasr_mfcc_set: unknown key %d", key);
  if (key == 4) {
    p->bd_fuse_dct = value != 0;
    return LIPASR_OK;
  }
  if (key == 3) {
    LP_CHECK_ARG(value >= 4 && value <= 4096 && (value & 3) == 0, "lipasr_mfcc_set: frames per workgroup %d (a multiple of 4)", value);
    p->bd_seg = value;
    return LIPASR_OK;
  }
  if (key == 2) {
    p->prefer_fused = value != 0;
    return LIPASR_OK;
  }
  if (key == 1) {
    LP_CHECK_ARG(value >= 1 && value <= 4096, "lipasr_mfcc_set: resampler workgroup target %d", value);
    p->rs_target_wgs = value;
    return LIPASR_OK;
  }
  p->stage_mask = value;
  return LIPASR_OK;
}

}  // namespace lipasr

extern "C" {

// ---------------------------------------------------------------- plan objects
int lipasr_mfcc_create(lipasr_handle_t h, int sr_in, int n_samp_max, int batch_max, int n_fft, int hop, lipasr_mfcc_t* out) {
  LP_CHECK_ARG(out != nullptr, "lipasr_mfcc_create: out is null");
  MfccPlan* p = nullptr;
  int rc = plan_build(h, sr_in, n_samp_max, batch_max, n_fft, hop, &p);
  if (rc != LIPASR_OK) return rc;
  *out = static_cast<lipasr_mfcc*>(p);
  return LIPASR_OK;
}

int lipasr_mfcc_destroy(lipasr_mfcc_t p) {
  LP_CHECK_ARG(p != nullptr, "lipasr_mfcc_destroy: null plan");
  DeviceGuard g(p->ctx->device);
  plan_unregister(p);
  mfcc_plan_free(p);
  return LIPASR_OK;
}

int lipasr_mfcc_plan_dims(lipasr_mfcc_t p, int* n_y, int* n_frames, int* fused) {
  LP_CHECK_ARG(p && n_y && n_frames, "lipasr_mfcc_plan_dims: null argument");
  *n_y = p->n_y;
  *n_frames = p->n_frames;
  if (fused) *fused = p->fused ? 1 : 0;
  return LIPASR_OK;
}

int lipasr_mfcc_extract(lipasr_mfcc_t p, const void* wav, int sample_format, const int* n_valid, int batch, int utterance_length,
                        const double* affine_mean, const double* affine_scale, float* out, lipasr_stream_t stream) {
  int rc = plan_check("lipasr_mfcc_extract", p, batch, utterance_length);
  if (rc != LIPASR_OK) return rc;
  return plan_run(p, wav, sample_format, n_valid, batch, utterance_length, affine_mean, affine_scale, out, S(stream));
}

int lipasr_mfcc_plan_resample(lipasr_mfcc_t p, const float* wav, int batch, float* y, lipasr_stream_t stream) {
  int rc = plan_check("lipasr_mfcc_plan_resample", p, batch, 1);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(wav && y, "lipasr_mfcc_plan_resample: null argument");
  return plan_resample(p, wav, batch, y, S(stream));
}

int lipasr_mfcc_plan_from_22k(lipasr_mfcc_t p, const float* y, int batch, int n_y, int utterance_length, const double* affine_mean,
                              const double* affine_scale, float* out, lipasr_stream_t stream) {
  int rc = plan_check("lipasr_mfcc_plan_from_22k", p, batch, utterance_length);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(y && out, "lipasr_mfcc_plan_from_22k: null argument");
  return plan_from_22k(p, y, batch, n_y, utterance_length, affine_mean, affine_scale, out, S(stream));
}

int lipasr_mfcc_plan_profile_begin(lipasr_mfcc_t p, int max_calls) { return plan_profile_begin(p, max_calls); }
int lipasr_mfcc_plan_profile_end(lipasr_mfcc_t p, float* avg_ms3, int* n_calls) { return plan_profile_end(p, avg_ms3, n_calls); }
int lipasr_mfcc_plan_set(lipasr_mfcc_t p, int key, int value) { return plan_set(p, key, value); }

// ---------------------------------------------------------------- the handle's default plan (round-1/2 entry points)
int lipasr_mfcc_plan(lipasr_handle_t h, int sr_in, int n_samp, int batch_max) {
  return lipasr_mfcc_plan_ex(h, sr_in, n_samp, batch_max, kNFft, kHop);
}

int lipasr_mfcc_plan_ex(lipasr_handle_t h, int sr_in, int n_samp, int batch_max, int n_fft, int hop) {
  LP_CHECK_ARG(h != nullptr, "lipasr_mfcc_plan: null handle");
  MfccPlan* p = nullptr;
  int rc = plan_build(h, sr_in, n_samp, batch_max, n_fft, hop, &p);
  if (rc != LIPASR_OK) return rc;
  if (h->mfcc) {
    DeviceGuard g(h->device);
    MfccPlan* old = h->mfcc;
    plan_unregister(old);
    mfcc_plan_free(old);
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
  return plan_check(fn, h->mfcc, batch, L);
}

int lipasr_resample_f32(lipasr_handle_t h, const float* wav, int batch, float* y, lipasr_stream_t stream) {
  int rc = mfcc_check("lipasr_resample_f32", h, batch, 1);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(wav && y, "lipasr_resample_f32: null argument");
  return plan_resample(h->mfcc, wav, batch, y, S(stream));
}

int lipasr_mfcc_from_22k(lipasr_handle_t h, const float* y, int batch, int n_y, int utterance_length,
                         const double* affine_mean, const double* affine_scale, float* out, lipasr_stream_t stream) {
  int rc = mfcc_check("lipasr_mfcc_from_22k", h, batch, utterance_length);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(y && out, "lipasr_mfcc_from_22k: null argument");
  return plan_from_22k(h->mfcc, y, batch, n_y, utterance_length, affine_mean, affine_scale, out, S(stream));
}

int lipasr_mfcc_f32(lipasr_handle_t h, const float* wav, int batch, int utterance_length, const double* affine_mean,
                    const double* affine_scale, float* out, lipasr_stream_t stream) {
  int rc = mfcc_check("lipasr_mfcc_f32", h, batch, utterance_length);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(wav && out, "lipasr_mfcc_f32: null argument");
  LP_CHECK_ARG((affine_mean == nullptr) == (affine_scale == nullptr), "lipasr_mfcc_f32: give both affine arrays or neither");
  return plan_run(h->mfcc, wav, 0, nullptr, batch, utterance_length, affine_mean, affine_scale, out, S(stream));
}

int lipasr_mfcc_i16(lipasr_handle_t h, const int16_t* pcm, const int* n_valid, int batch, int utterance_length,
                    const double* affine_mean, const double* affine_scale, float* out, lipasr_stream_t stream) {
  int rc = mfcc_check("lipasr_mfcc_i16", h, batch, utterance_length);
  if (rc != LIPASR_OK) return rc;
  return plan_run(h->mfcc, pcm, 1, n_valid, batch, utterance_length, affine_mean, affine_scale, out, S(stream));
}

int lipasr_mfcc_profile_begin(lipasr_handle_t h, int max_calls) {
  LP_CHECK_ARG(h != nullptr, "lipasr_mfcc_profile_begin: bad argument");
  if (!h->mfcc) { set_error("lipasr_mfcc_profile_begin: call lipasr_mfcc_plan first"); return LIPASR_ESTATE; }
  return plan_profile_begin(h->mfcc, max_calls);
}

int lipasr_mfcc_profile_end(lipasr_handle_t h, float* avg_ms3, int* n_calls) {
  LP_CHECK_ARG(h && avg_ms3 && n_calls, "lipasr_mfcc_profile_end: null argument");
  if (!h->mfcc) { set_error("lipasr_mfcc_profile_end: call lipasr_mfcc_plan first"); return LIPASR_ESTATE; }
  return plan_profile_end(h->mfcc, avg_ms3, n_calls);
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

/* Knobs of the handle's default plan (lipasr_mfcc_plan_set is the per-plan form).  key 0: stage mask for profiling and
 * A/B runs (bit0 skip the FFT passes, bit1 skip the mel reduction -- wrong results by design; bit2 VALU resampler; bit7 =
 * 128: the three-kernel path instead of the fused resample -> STFT kernel).  key 1: workgroups the persistent resampler of
 * the three-kernel path aims for. */
int lipasr_debug_set(lipasr_handle_t h, int key, int value) {
  LP_CHECK_ARG(h != nullptr, "lipasr_debug_set: null handle");
  LP_CHECK_ARG(key >= 0 && key <= 2, "lipasr_debug_set: unknown key %d", key);
  if (!h->mfcc) { set_error("lipasr_debug_set: call lipasr_mfcc_plan first"); return LIPASR_ESTATE; }
  int rc = plan_set(h->mfcc, key, value);
  if (rc == LIPASR_OK && key == 1) h->rs_target_wgs = value;  // kept in the handle: a later lipasr_mfcc_plan inherits it
  return rc;
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
    case 3: case 4: case 5: case 6: case 7: {
      LP_CHECK_ARG(sr_in >= 1000 && sr_in != kSr, "lipasr_debug_table: sr_in=%d", sr_in);
      Polyphase pp = build_polyphase(sr_in, kSr);
      if (which == 3) v = pp.h;
      else if (which == 4) v = {(float)pp.up, (float)pp.down, (float)pp.taps, (float)pp.left};
      else if (which == 5) v.assign(pp.n_off.begin(), pp.n_off.end());
      else {
        std::vector<float> hb;
        std::vector<int> lo;
        if (!build_band_tables(pp, &hb, &lo)) { set_error("lipasr_debug_table: no banded form for sr_in=%d", sr_in); return LIPASR_EUNSUPPORTED; }
        if (which == 6) v = hb;
        else v.assign(lo.begin(), lo.end());
      }
      break;
    }
    case 8: {  // the mel bank as the kernel applies it (pair form), expanded to dense [128*1025]
      MelPairs mp = mel_pairs();
      LP_CHECK_ARG(mp.ok, "lipasr_debug_table: mel bank has no pair form");
      v.assign((size_t)kNMels * kNBins, 0.0f);
      for (int m = 0; m < kNMels; ++m)
        for (int i = 0; i < mp.len[m]; ++i) {
          const int b = mp.start[m] + i;
          v[(size_t)m * kNBins + b] += mp.wlo[b];
          if (m + 1 < kNMels) v[(size_t)(m + 1) * kNBins + b] += mp.whi[b];
        }
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
