// Shared by the STFT kernels of K1 (mfcc.hip: Stockham kernels; stft_bdft.hip: block-DFT kernel on the matrix pipe).
#pragma once
#include "common.h"
#include "mfcc_tables.h"

namespace lipasr {

struct StftArgs {
  const float* y;  // [B][n_y]
  int n_y, n_frames;
  const float* hann;
  const float2* tw;
  const float* mel_wlo;  // [1025]
  const float* mel_whi;  // [1025]
  const int* mel_start;  // [128] run of bins whose lower filter is m
  const int* mel_len;
  float* db;    // [B][n_frames][128]
  float* fmax;  // [B][n_frames]
  int stage_mask;  // profiling only: bit0 skip the FFT passes, bit1 skip the mel reduction (results are wrong)
  // clips of different lengths in one launch (stft_mel2_kernel, stft_bdft_kernel): samples per clip, or null; n_y / n_frames
  // above are then the longest clip's (the strides of y, db, fmax) and every clip uses its own
  const int* n_valid;
  int sr_in, n_samp_max;
};

// per-clip lengths, the expressions of tables::resampled_lengths (librosa.load -> resampy int(n ratio), fix_length ceil)
__device__ __forceinline__ void clip_lengths(int n, int sr_in, int* n_vy, int* n_y, int* n_frames) {
  const double r = (double)tables::kSr / (double)sr_in;
  const double v = (double)n * r;
  *n_vy = (int)v;
  *n_y = (int)ceil(v);
  *n_frames = (*n_y >= 2) ? 1 + *n_y / tables::kHop : 0;
}

// np.pad(y, 1024, mode='reflect') index: position j relative to y[0], any j, n >= 2
__device__ __forceinline__ int reflect_index(int j, int n) {
  if ((unsigned)j < (unsigned)n) return j;  // interior frames never reflect
  const int period = 2 * (n - 1);
  int m = j % period;
  if (m < 0) m += period;
  return m < n ? m : period - m;
}

// the same for a clip longer than the padding (n > 2048 >= any |overshoot|): one reflection, no division
__device__ __forceinline__ int reflect_once(int j, int n) {
  const int lo = j < 0 ? -j : j;
  return lo < n ? lo : 2 * (n - 1) - lo;
}

__device__ __forceinline__ void lds_barrier2() {
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS traffic only (the kernel's global stores need no ordering)
  __builtin_amdgcn_s_barrier();
}

// ---- stft_bdft.hip: constant tables of the block-DFT kernel (device pointers, owned by the MFCC plan) ----
struct BdftTables {
  uint4* cfrag = nullptr;    // [4 tiles][2 planes][64 lanes]: stage-1 matrix (cos | -sin of W64), fp16 hi / lo fragments
  uint4* efrag = nullptr;    // [4 k-steps][2 planes][64 lanes]: stage-2 matrix (W32, complex as real 64 x 32)
  float4* tw = nullptr;      // [2 tiles][8][64 lanes]: inter-stage twiddles W2048^(n2 k1) x 2^-10 in accumulator order
  float* wlo = nullptr;      // [1024] mel weights of the lower / upper filter of every bin, x 0.25 / 8192^2
  float* whi = nullptr;
  unsigned long long* smask = nullptr;  // [16]: lanes whose bin 16 lane + j opens a mel run
  int4* mpos = nullptr;      // [128][2]: staging positions summed into mel m (run m of the lower plane | run m - 1 of the upper)
};
int bdft_tables_build(BdftTables* t);
void bdft_tables_free(BdftTables* t);
// `dct`: when non-null and one workgroup covers a whole clip (seg_frames >= n_frames <= 64, L <= 64), the kernel also applies the
// top_db floor and the DCT (+ optional StandardScaler affine) and writes the features: no dct_kernel launch afterwards.
struct BdftDct {
  int L = 0;                        // utterance_length: output frames per clip
  const float4* dct_frag = nullptr; // DCT-II rows in MFMA fragment order (MfccPlan::d_dct)
  const double* aff_mean = nullptr;
  const double* aff_scale = nullptr;
  float* out = nullptr;             // [batch][20 * L]
};
bool bdft_can_fuse_dct(int n_frames, int seg_frames, int L);
int launch_stft_bdft(const StftArgs& a, const BdftTables& t, int batch, int seg_frames, const BdftDct* dct, hipStream_t st);

}  // namespace lipasr
