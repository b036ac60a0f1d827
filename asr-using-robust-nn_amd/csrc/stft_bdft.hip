// K1 stage 2 on the matrix pipe: STFT 2048/512 + mel + dB as a BLOCK DFT (gfx950).
//
// Replaces the arithmetic of librosa.feature.mfcc's STFT (extract_features_construct_dataset.py:30: n_fft 2048, hop 512,
// periodic Hann, reflect padding) behind the same interface as stft_mel2_kernel (mel dB [B][frames][128] + per-frame maximum).
//
// 1. Frames overlap by 3/4, so the padded clip is cut into BLOCKS of 512 samples and each block is transformed ONCE:
//        B_j[k] = sum_{m<512} ypad[512 j + m] e^(-2 pi i m k / 2048),      k = 0 .. 1024
//    and a frame's (rectangular-window) spectrum is four block spectra with trivial twiddles,
//        X_f[k] = sum_{b<4} (-i)^(b k) B_{f+b}[k].
// 2. The periodic Hann window is 0.5 - 0.25 e^(+2 pi i n/N) - 0.25 e^(-2 pi i n/N): in the frequency domain exactly
//        Xw[k] = 0.5 X[k] - 0.25 (X[k-1] + X[k+1]).
// 3. The block transform is two matrix stages, m = 32 n1 + n2, k = k1 + 64 k2:
//        A[n2][k1]     = sum_{n1<16} y[32 n1 + n2] W64^(n1 k1)            32 x 16 data  x  16 x (64 cos | 64 sin)
//        A'[n2][k1]    = A[n2][k1] W2048^(n2 k1)                          fp32, vector ALU
//        B[k1 + 64 k2] = sum_{n2<32} A'[n2][k1] W32^(n2 k2),  k2 < 16     64 x (32 re | 32 im)  x  64 x (16 re | 16 im)
//    both on v_mfma_f32_32x32x16_f16 with every operand split into two fp16 planes (hi = round to 11 bits, lo = fp16 of the
//    remainder) and three of the four cross terms accumulated in fp32 -- the resampler's technique: products of fp16 numbers
//    are exact in fp32, the error is the 2^-22 of the representations.  The accumulator layout of stage 1 (lane = k1, registers
//    = n2) IS the A-operand layout of stage 2 with the contraction index permuted, and the constant matrix of stage 2 is stored
//    in that permuted order, so nothing is transposed between the stages.  36 matrix instructions per block of 512 samples.
//
// One workgroup = one clip (or a run of `seg_frames` of its frames), four wavefronts, two workgroups per CU.  Per iteration
// every wavefront transforms one block into a ring of 7 block spectra in LDS (planar re | im, XOR-swizzled so that the
// strided accumulator stores and the unit-stride reads are both conflict-free), then finishes one frame: lane l owns bins
// 16 l .. 16 l + 15, reads its four block spectra, combines, applies the Hann taps (neighbour bins by DPP wave shifts), squares,
// and reduces the two mel weights of each bin by a SEGMENTED SCAN in registers (the mel runs that open inside a lane's 16
// bins are wave-uniform lane masks in SGPRs); the scan values go to a wave-private staging slot (the ring slot this
// wavefront overwrites next) and lane l sums the <= 6 segment ends of mel l and mel l + 64.  Two LDS-only barriers per
// iteration (stft_mel2_kernel: eleven per frame quad).
#include "stft.h"

namespace lipasr {

using namespace tables;

typedef _Float16 bd_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 bd_h2 __attribute__((ext_vector_type(2)));
typedef float bd_f2 __attribute__((ext_vector_type(2)));
typedef float bd_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBdPlane = 1028;            // floats per plane of a ring slot: bins 0 .. 1024, padded to a multiple of 4
constexpr int kBdSlot = 2 * kBdPlane;     // re | im
constexpr int kBdSlots = 7;               // blocks f .. f + 6 serve the four frames of an iteration
constexpr int kBdTwF4 = 16 * 64;          // float4 entries of the twiddle table
constexpr int kBdLdsFloats = kBdSlots * kBdSlot + 4 * kBdTwF4;  // 73 952 bytes: two workgroups per CU
constexpr int kBdDummy = 2048;            // staging position that always reads 0
constexpr double kBdSig = 2048.0, kBdTap = 64.0, kBdMid = 1.0 / 1024.0;
constexpr float kBdOut = 8192.0f;         // accumulator units per signal unit: kBdSig * kBdTap * kBdMid * kBdTap

struct BdftArgs {
  StftArgs st;
  const uint4* cfrag;
  const uint4* efrag;
  const float4* tw;
  const float* wlo;
  const float* whi;
  const unsigned long long* smask;
  const int4* mpos;
  int seg_frames;
  // fused top_db floor + DCT epilogue (one workgroup per clip); L = 0: off
  int L;
  const float4* dct_frag;
  const double* aff_mean;
  const double* aff_scale;
  float* out;
};

// Eight fp32 values -> two fp16 planes as matrix operands: hi = RNE(x), lo = RNE(x - hi) (v_fma_mix: the subtraction reads hi as
// fp16, no conversion back).  The low plane is ONE asm statement that ends in s_nop 1: on gfx90a+ a vector-ALU write of a
// register needs two wait states before a matrix instruction reads it, and the compiler's hazard recogniser does not look
// inside inline asm (a build whose scheduler put the matrix instruction right behind the last v_fma_mixhi computed garbage).
__device__ __forceinline__ void bd_split8(const float (&x)[8], bd_h8& hi, bd_h8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector((bd_f2){x[2 * i], x[2 * i + 1]}, bd_h2));
  asm("v_fma_mixlo_f16 %0, %4, 1.0, -%12 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %1, %6, 1.0, -%13 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %2, %8, 1.0, -%14 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %10, 1.0, -%15 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %5, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %7, 1.0, -%13 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %9, 1.0, -%14 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %11, 1.0, -%15 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "s_nop 1"
      : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
  hi = __builtin_bit_cast(bd_h8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bd_h8, make_uint4(l[0], l[1], l[2], l[3]));
}

__device__ __forceinline__ bd_h8 bd_pack(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_bit_cast(bd_h8, make_uint4(a, b, c, d));
}

// lane i <- lane i - 1 (lane 0 <- 0) / lane i <- lane i + 1 (lane 63 <- 0)
__device__ __forceinline__ float bd_from_below(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));  // wave_shr:1
}
__device__ __forceinline__ float bd_from_above(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));  // wave_shl:1
}

// maximum over the wavefront by DPP (no LDS round trips): quads, half rows, rows, then row broadcasts; wave-uniform result
__device__ __forceinline__ float bd_wave_max(float x) {
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x141, 0xf, 0xf, false)));  // row_half_mirror
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x140, 0xf, 0xf, false)));  // row_mirror
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x142, 0xa, 0xf, false)));  // row_bcast:15
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x143, 0xc, 0xf, false)));  // row_bcast:31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// the 8 samples of lane (li, h) in block jg: y index 512 jg - 1024 + 32 (8 h + i) + li, reflect-padded, zeros from n_vy on
__device__ __forceinline__ void bd_fetch(const float* __restrict__ yu, int jg, int li, int h, int n_y, int n_vy, float (&ys)[8]) {
  const int i0 = 512 * jg - 1024;    // y index of the block's first sample
  const int p0 = i0 + 256 * h + li;
  if (i0 >= 0 && i0 + 512 <= n_vy) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ys[i] = yu[p0 + 32 * i];
  } else if (n_y > kNFft) {  // clips longer than the padding reflect once
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = reflect_once(p0 + 32 * i, n_y);
      ys[i] = (k < n_vy) ? yu[k] : 0.0f;  // [n_vy, n_y): fix_length's zeros
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = reflect_index(p0 + 32 * i, n_y);
      ys[i] = (k < n_vy) ? yu[k] : 0.0f;
    }
  }
}

struct BdConst {
  bd_h8 ch[4], cl[4], eh[4], el[4];  // constant operands of the two matrix stages: 64 registers
};

__device__ __forceinline__ void bd_load_const(const BdftArgs& a, int lane, BdConst& K) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    K.ch[t] = __builtin_bit_cast(bd_h8, a.cfrag[(2 * t + 0) * 64 + lane]);
    K.cl[t] = __builtin_bit_cast(bd_h8, a.cfrag[(2 * t + 1) * 64 + lane]);
    K.eh[t] = __builtin_bit_cast(bd_h8, a.efrag[(2 * t + 0) * 64 + lane]);
    K.el[t] = __builtin_bit_cast(bd_h8, a.efrag[(2 * t + 1) * 64 + lane]);
  }
}

// the staging positions summed into mel `lane` and mel `lane + 64`: 12 positions, two per register
__device__ __forceinline__ void bd_load_mpos(const BdftArgs& a, int lane, unsigned (&mp)[6]) {
  const int4 pa0 = a.mpos[2 * lane], pb0 = a.mpos[2 * lane + 1], pa1 = a.mpos[2 * (lane + 64)], pb1 = a.mpos[2 * (lane + 64) + 1];
  mp[0] = (unsigned)pa0.x | ((unsigned)pa0.y << 16); mp[1] = (unsigned)pa0.z | ((unsigned)pb0.x << 16); mp[2] = (unsigned)pb0.y | ((unsigned)pb0.z << 16);
  mp[3] = (unsigned)pa1.x | ((unsigned)pa1.y << 16); mp[4] = (unsigned)pa1.z | ((unsigned)pb1.x << 16); mp[5] = (unsigned)pb1.y | ((unsigned)pb1.z << 16);
}

// the lane's 8 samples -> the two fp16 planes of the first stage's data operand
__device__ __forceinline__ void bd_data_planes(const float (&ys)[8], bd_h8& ah, bd_h8& al) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = fminf(fmaxf(ys[i] * (float)kBdSig, -65000.0f), 65000.0f);
  bd_split8(v, ah, al);
}

// One block of 512 samples (data planes ah / al) -> its spectrum, bins 0 .. 1024, into the ring slot `slot_base`
__device__ __forceinline__ void bd_block(const bd_h8& ah, const bd_h8& al, const BdConst& K, const float4* __restrict__ twl,
                                         float* __restrict__ slot_base, int lane) {
  const int li = lane & 31, h = lane >> 5, k2 = li & 15, plane = li >> 4;
  float* slot = slot_base + plane * kBdPlane;
  float b1024 = 0.0f;
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    // the tile's 8 twiddle quads leave for the registers before its first-stage matrix chain (they land behind it)
    float4 tw8[8];
#pragma unroll
    for (int ep = 0; ep < 8; ++ep) tw8[ep] = twl[(t2 * 8 + ep) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    // stage 1: tiles cos (re) and -sin (im) of k1 = 32 t2 + li; rows n2, K = n1
    bd_f32x16 a1[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int t = 2 * c + t2;
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[c][e] = 0.0f;
      a1[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, K.ch[t], a1[c], 0, 0, 0);
      a1[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, K.cl[t], a1[c], 0, 0, 0);
      a1[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, K.ch[t], a1[c], 0, 0, 0);
    }
    if (t2 == 0) {
      // bin 1024 (k1 = 0 with k2 = 16: outside the second stage's 16 columns) = sum_m y[m] (-1)^m = sum_n2 (-1)^n2 A[n2][0]:
      // column 0 of the first cosine tile, i.e. lanes 0 and 32, register parity = n2 parity
      float s8 = 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) s8 += (e & 1) ? -a1[0][e] : a1[0][e];
      b1024 = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(s8), 0)) +
               __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s8), 32))) * (float)(kBdMid * kBdTap);
    }
    // inter-stage twiddle, split into planes, second stage.  Register e of this lane is n2 = (e & 3) + 8 (e >> 2) + 4 h,
    // k1 = 32 t2 + li; the second stage contracts over (re | im) x n2 in accumulator order: k-step s = 2 c + half takes
    // component c of registers 8 half .. 8 half + 7; columns (16 re | 16 im) of k2
    bd_f32x16 a3;
#pragma unroll
    for (int e = 0; e < 16; ++e) a3[e] = 0.0f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float tr[8], tq[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ep = 4 * half + q;
        const float4 t = tw8[ep];
        const float re0 = a1[0][2 * ep], im0 = a1[1][2 * ep], re1 = a1[0][2 * ep + 1], im1 = a1[1][2 * ep + 1];
        tr[2 * q] = re0 * t.x - im0 * t.y; tq[2 * q] = re0 * t.y + im0 * t.x;
        tr[2 * q + 1] = re1 * t.z - im1 * t.w; tq[2 * q + 1] = re1 * t.w + im1 * t.z;
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int s = 2 * c + half;
        bd_h8 oh, ol;
        bd_split8(c ? tq : tr, oh, ol);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh, K.eh[s], a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh, K.el[s], a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ol, K.eh[s], a3, 0, 0, 0);
      }
    }
    // lane (column li = (plane, k2), h): registers 4 b .. 4 b + 3 are bins k = 64 k2 + 32 t2 + 8 b + 4 h + (0 .. 3)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int k = 64 * k2 + 32 * t2 + 8 * b + 4 * h;
      *reinterpret_cast<float4*>(slot + (k ^ ((k2 & 7) << 2))) = make_float4(a3[4 * b], a3[4 * b + 1], a3[4 * b + 2], a3[4 * b + 3]);
    }
  }
  if (lane == 0) slot[1024] = b1024;
  if (lane == 16) slot[1024] = 0.0f;  // (plane 1)
}

// One frame: its four block spectra sb[0 .. 3] -> combine -> Hann taps -> power -> segmented scans of the two mel weight planes
// (wl / wh: the lane's 16 + 16 mel weights)
__device__ __forceinline__ void bd_frame_scan(const float* const (&sb)[4], const float (&wl)[16], const float (&wh)[16],
                                              const unsigned long long (&sm)[16], int lane, float (&sl)[16], float (&sh)[16]) {
  float xr[16], xi[16];
  // the four block spectra of bins 16 l + 4 q .. + 3, one quad ahead of the arithmetic (two sets of 8 float4 in flight:
  // hoisting all four quads together would take 128 registers, none ahead exposes four LDS round trips)
  float4 br[2][4], bi[2][4];
  auto fetch_quad = [&](int q, float4 (&r)[4], float4 (&i)[4]) {
    const int k = 16 * lane + 4 * q;
    const int pos = k ^ (((k >> 6) & 7) << 2);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      r[b] = *reinterpret_cast<const float4*>(sb[b] + pos);
      i[b] = *reinterpret_cast<const float4*>(sb[b] + kBdPlane + pos);
    }
  };
  fetch_quad(0, br[0], bi[0]);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q < 3) fetch_quad(q + 1, br[(q + 1) & 1], bi[(q + 1) & 1]);
    const float4 (&r)[4] = br[q & 1];
    const float4 (&i)[4] = bi[q & 1];
    // X = B0 + w B1 + w^2 B2 + w^3 B3, w = (-i)^k, k mod 4 = the component
    xr[4 * q + 0] = (r[0].x + r[2].x) + (r[1].x + r[3].x);
    xi[4 * q + 0] = (i[0].x + i[2].x) + (i[1].x + i[3].x);
    xr[4 * q + 1] = (r[0].y - r[2].y) + (i[1].y - i[3].y);
    xi[4 * q + 1] = (i[0].y - i[2].y) - (r[1].y - r[3].y);
    xr[4 * q + 2] = (r[0].z + r[2].z) - (r[1].z + r[3].z);
    xi[4 * q + 2] = (i[0].z + i[2].z) - (i[1].z + i[3].z);
    xr[4 * q + 3] = (r[0].w - r[2].w) - (i[1].w - i[3].w);
    xi[4 * q + 3] = (i[0].w - i[2].w) + (r[1].w - r[3].w);
    __builtin_amdgcn_sched_barrier(0);  // (pins the order: quad q + 1's reads are issued before quad q's arithmetic)
  }
  // neighbours across the lane boundary: X[16 l - 1] from the lane below (lane 0: X[-1] = conj X[1]), X[16 l + 16] from the
  // lane above (lane 63: X[1024] = sum of the four blocks' bin 1024, real)
  float lr = bd_from_below(xr[15]), lq = bd_from_below(xi[15]);
  float ur = bd_from_above(xr[0]), uq = bd_from_above(xi[0]);
  const float x1024 = (sb[0][1024] + sb[2][1024]) + (sb[1][1024] + sb[3][1024]);
  if (lane == 0) { lr = xr[1]; lq = -xi[1]; }
  if (lane == 63) { ur = x1024; uq = 0.0f; }
  float pw[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float mr = (j ? xr[j - 1] : lr) + (j < 15 ? xr[j + 1] : ur);
    const float mq = (j ? xi[j - 1] : lq) + (j < 15 ? xi[j + 1] : uq);
    const float hr = xr[j] - 0.5f * mr, hq = xi[j] - 0.5f * mq;  // 2 Xw (the 1/4 of the power is in the weights)
    pw[j] = hr * hr + hq * hq;
  }
  // segmented scan of the weighted powers over the lane's 16 bins: a sum restarts where a mel run opens
  float cl_ = 0.0f, ch_ = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float pl_, ph_;
    if (j == 0) {
      pl_ = 0.0f; ph_ = 0.0f;
    } else {
      asm("v_cndmask_b32 %0, %1, 0, %2" : "=v"(pl_) : "v"(cl_), "s"(sm[j]));
      asm("v_cndmask_b32 %0, %1, 0, %2" : "=v"(ph_) : "v"(ch_), "s"(sm[j]));
    }
    cl_ = fmaf(wl[j], pw[j], pl_);
    ch_ = fmaf(wh[j], pw[j], ph_);
    sl[j] = cl_; sh[j] = ch_;
  }
}

__device__ __forceinline__ void bd_load_weights(const BdftArgs& a, int lane, float (&wl)[16], float (&wh)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 l4 = reinterpret_cast<const float4*>(a.wlo)[4 * lane + q], h4 = reinterpret_cast<const float4*>(a.whi)[4 * lane + q];
    wl[4 * q] = l4.x; wl[4 * q + 1] = l4.y; wl[4 * q + 2] = l4.z; wl[4 * q + 3] = l4.w;
    wh[4 * q] = h4.x; wh[4 * q + 1] = h4.y; wh[4 * q + 2] = h4.z; wh[4 * q + 3] = h4.w;
  }
}

// scans -> staging `stg` (2056 floats no other wavefront touches now) -> mel `lane` and `lane + 64` -> dB -> a.st.db; returns the
// frame's maximum (wave-uniform; lane 0 also stores it to a.st.fmax)
__device__ __forceinline__ float bd_frame_finish(const BdftArgs& a, float* __restrict__ stg, const float (&sl)[16], const float (&sh)[16],
                                                 const unsigned (&mp)[6], int lane, int u, int f) {
  const int x = (lane >> 2) & 3;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    *reinterpret_cast<float4*>(stg + 16 * lane + 4 * (q ^ x)) = make_float4(sl[4 * q], sl[4 * q + 1], sl[4 * q + 2], sl[4 * q + 3]);
    *reinterpret_cast<float4*>(stg + 1024 + 16 * lane + 4 * (q ^ x)) = make_float4(sh[4 * q], sh[4 * q + 1], sh[4 * q + 2], sh[4 * q + 3]);
  }
  if (lane == 0) stg[kBdDummy] = 0.0f;
  const float m0 = ((stg[mp[0] & 0xffff] + stg[mp[0] >> 16]) + stg[mp[1] & 0xffff]) + ((stg[mp[1] >> 16] + stg[mp[2] & 0xffff]) + stg[mp[2] >> 16]);
  const float m1 = ((stg[mp[3] & 0xffff] + stg[mp[3] >> 16]) + stg[mp[4] & 0xffff]) + ((stg[mp[4] >> 16] + stg[mp[5] & 0xffff]) + stg[mp[5] >> 16]);
  const float d0 = 10.0f * log10f(fmaxf(1e-10f, m0)), d1 = 10.0f * log10f(fmaxf(1e-10f, m1));
  float* dbp = a.st.db + ((size_t)u * a.st.n_frames + f) * 128;
  dbp[lane] = d0;
  dbp[lane + 64] = d1;
  const float mx = bd_wave_max(fmaxf(d0, d1));
  if (lane == 0) a.st.fmax[(size_t)u * a.st.n_frames + f] = mx;
  return mx;
}

// The optional fused epilogue: dct_kernel's arithmetic, instruction for instruction (mfcc.hip), on this workgroup's own dB tile:
// top_db floor against the clip maximum, DCT-II as a 32 x 32 x 128 contraction per wavefront on v_mfma_f32_32x32x2_f32
// (wavefronts 0 and 1: 32 frames each), optional affine in fp64.  The tile comes back from L2 (this workgroup stored it); the
// ring is free by now and holds the transposed image.  NT = threads of the workgroup; red: one float per wavefront.
template <int NT>
__device__ __forceinline__ void bd_dct_epilogue(const BdftArgs& a, float* __restrict__ dbs, float* __restrict__ red, int tid, int u, int nf, float clip_max) {
  const int lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int L = a.L;
  const int chunk = min(64, (L + 3) & ~3), tp = chunk + 1;
  const int tl = min(chunk, L), tu = max(0, min(nf, tl));
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wavefront's dB stores have left
  lds_barrier2();                      // every wavefront is done with ring and staging
  if (lane == 0) red[wave] = clip_max;
  __syncthreads();
  float mxall = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) mxall = fmaxf(mxall, red[w]);
  const float thr = mxall - 80.0f;  // top_db = 80
  const float* src = a.st.db + (size_t)u * a.st.n_frames * 128;
  const int n_live = tu * 128;
  constexpr int NJ = 64 * 128 / NT;
  float stage[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = tid + NT * j;  // frame i / 128, mel i % 128
    stage[j] = (i < n_live) ? __builtin_nontemporal_load(src + i) : 0.0f;
  }
  float4 av4[16];
  const float4* ap = a.dct_frag + (li * 2 + h) * 16;
  if (wave < 2) {
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) av4[s4] = ap[s4];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = tid + NT * j, t = i >> 7;
    if (t < chunk) dbs[(i & 127) * tp + t] = (i < n_live) ? fmaxf(stage[j], thr) : 0.0f;
  }
  __syncthreads();
  if (wave < 2) {
    bd_f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    const float* bp = dbs + h * tp + min(wave * 32 + li, chunk - 1);
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      const float a4[4] = {av4[s4].x, av4[s4].y, av4[s4].z, av4[s4].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], bp[2 * (4 * s4 + e) * tp], acc, 0, 0, 0);
    }
    const int t = wave * 32 + li;
    const int n_out = kNMfcc * L;
    if (t < tl) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int c = (q & 3) + 8 * (q >> 2) + 4 * h;
        if (c < kNMfcc) {
          float v = (t < tu) ? acc[q] : 0.0f;
          const int oo = c * L + t;
          if (a.aff_mean) v = (float)(((double)v - a.aff_mean[oo]) / a.aff_scale[oo]);
          a.out[(size_t)u * n_out + oo] = v;
        }
      }
    }
  }
}

// this clip's own lengths (per-clip lengths: frame count and reflect padding follow them)
__device__ __forceinline__ void bd_clip(const BdftArgs& a, int u, int& n_y, int& n_vy, int& n_frames) {
  n_y = a.st.n_y; n_frames = a.st.n_frames; n_vy = a.st.n_y;
  if (a.st.n_valid) {
    clip_lengths(min(max(a.st.n_valid[u], 0), a.st.n_samp_max), a.st.sr_in, &n_vy, &n_y, &n_frames);
    n_frames = min(n_frames, a.st.n_frames);
  }
}

// ---------------------------------------------------------------------------------------------
// The kernel: four wavefronts, two workgroups per CU; every wavefront alternates between one block and one frame; ring of 7
// block spectra; the staging slot of a wavefront is the ring slot it overwrites next.  Two LDS-only barriers per iteration.
// (Measured and not kept, round 4: a split-role form -- eight wavefronts, four that only transform blocks and four that only
// finish frames one iteration behind, ring of 11, one barrier per iteration -- 130 us against 121: its phases alone cost
// 27 us skeleton + 77 us blocks + 52 us frames and overlap only halfway)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void stft_bdft_kernel(BdftArgs a) {
  extern __shared__ __attribute__((aligned(16))) float bd_lds[];
  float* ring = bd_lds;
  float4* twl = reinterpret_cast<float4*>(bd_lds + kBdSlots * kBdSlot);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int u = blockIdx.y;
  int n_y, n_vy, n_frames;
  bd_clip(a, u, n_y, n_vy, n_frames);
  const int F0 = blockIdx.x * a.seg_frames;
  // (workgroup-uniform, before any barrier.  With the fused epilogue a clip without a frame still gets its zero columns,
  // fix_frames' padding, extract_features_construct_dataset.py:33-37)
  if (F0 >= n_frames && a.L <= 0) return;
  const int F1 = min(F0 + a.seg_frames, n_frames);
  BdConst K;
  bd_load_const(a, lane, K);
  unsigned long long sm[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) sm[j] = a.smask[j];
#pragma unroll
  for (int i = 0; i < kBdTwF4 / 256; ++i) twl[tid + 256 * i] = a.tw[tid + 256 * i];
  __syncthreads();
  const float* yu = a.st.y + (size_t)u * a.st.n_y;
  const int n_iter = (F0 < n_frames) ? (F1 - F0 + 3) >> 2 : -1;  // -1: no frame, the loop below does not run
  unsigned mp[6];
  bd_load_mpos(a, lane, mp);
  float clip_max = -INFINITY;  // over this wavefront's frames (fused epilogue)
  float ys[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ys[i] = 0.0f;
  if (wave >= 1 && n_iter >= 0) bd_fetch(yu, F0 + wave - 1, li, h, n_y, n_vy, ys);  // iteration -1: blocks 0 .. 2 by wavefronts 1 .. 3
  for (int it = -1; it < n_iter; ++it) {
    // ---------------------------------------------------------------- one block per wavefront
    const int rel = 4 * it + wave + 3;  // relative to F0; iteration -1 fills blocks 0 .. 2 (wavefronts 1 .. 3)
    const int jg = F0 + rel;
    const bool do_block = rel >= 0 && jg <= F1 + 2;  // (wave-uniform)
    bd_h8 ah, al;
    if (do_block) bd_data_planes(ys, ah, al);
    // the NEXT block's samples leave now: an HBM round trip that has a whole iteration to come back
    if (jg + 4 <= F1 + 2) bd_fetch(yu, jg + 4, li, h, n_y, n_vy, ys);
    if (do_block) bd_block(ah, al, K, twl, ring + (rel % kBdSlots) * kBdSlot, lane);
    // ---------------------------------------------------------------- one frame per wavefront
    const int fr = 4 * it + wave, f = F0 + fr;
    const bool do_frame = it >= 0 && f < F1;
    lds_barrier2();
    float sl[16], sh[16];
    if (do_frame) {
      float wl[16], wh[16];  // the lane's 32 mel weights (L1 / L2 resident), wanted after the combine
      bd_load_weights(a, lane, wl, wh);
      const float* sb[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) sb[b] = ring + ((fr + b) % kBdSlots) * kBdSlot;
      bd_frame_scan(sb, wl, wh, sm, lane, sl, sh);
    }
    lds_barrier2();  // every wavefront has read its four block spectra: blocks fr' .. fr' + 3 of this iteration are free
    if (do_frame) {
      // staging = the slot of block 4 it + wave, which is also the slot this wavefront's next block goes to
      const float mx = bd_frame_finish(a, ring + (fr % kBdSlots) * kBdSlot, sl, sh, mp, lane, u, f);
      clip_max = fmaxf(clip_max, mx);
    }
  }
  if (a.L <= 0) return;
  bd_dct_epilogue<256>(a, ring, ring + 128 * 65, tid, u, n_frames, clip_max);
}

// ---------------------------------------------------------------------------------------------
// host: constant tables (fp64 evaluation, exact angle reduction)
// ---------------------------------------------------------------------------------------------
static void bd_planes(const double (&v)[8], unsigned int* hi4, unsigned int* lo4) {
  unsigned short hi[8], lw[8];
  for (int j = 0; j < 8; ++j) {
    const _Float16 a = (_Float16)v[j];
    const _Float16 b = (_Float16)(v[j] - (double)a);
    memcpy(&hi[j], &a, 2);
    memcpy(&lw[j], &b, 2);
  }
  for (int w = 0; w < 4; ++w) {
    hi4[w] = (unsigned int)hi[2 * w] | ((unsigned int)hi[2 * w + 1] << 16);
    lo4[w] = (unsigned int)lw[2 * w] | ((unsigned int)lw[2 * w + 1] << 16);
  }
}

template <typename T>
static int bd_upload(T** dptr, const void* src, size_t bytes) {
  LP_HIP(hipMalloc(reinterpret_cast<void**>(dptr), bytes));
  LP_HIP(hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
  return LIPASR_OK;
}

void bdft_tables_free(BdftTables* t) {
  void* ptrs[] = {t->cfrag, t->efrag, t->tw, t->wlo, t->whi, t->smask, t->mpos};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  *t = BdftTables();
}

int bdft_tables_build(BdftTables* t) {
  auto cs = [](long num, long den, double* c, double* s) {  // cos / sin of 2 pi num / den, num reduced exactly
    const long r = ((num % den) + den) % den;
    const double ang = 2.0 * kPi * (double)r / (double)den;
    *c = cos(ang); *s = sin(ang);
  };
  // stage 1, B operand: tile = 2 comp + t2, lane (column li, half h) holds K = n1 = 8 h + i
  std::vector<unsigned int> cf((size_t)4 * 2 * 64 * 4), ef((size_t)4 * 2 * 64 * 4);
  for (int tile = 0; tile < 4; ++tile)
    for (int ln = 0; ln < 64; ++ln) {
      const int col = ln & 31, hh = ln >> 5, k1 = 32 * (tile & 1) + col;
      double v[8];
      for (int i = 0; i < 8; ++i) {
        double c, s;
        cs((long)(8 * hh + i) * k1, 64, &c, &s);
        v[i] = kBdTap * ((tile >> 1) ? -s : c);
      }
      bd_planes(v, &cf[((size_t)(2 * tile + 0) * 64 + ln) * 4], &cf[((size_t)(2 * tile + 1) * 64 + ln) * 4]);
    }
  // stage 2, B operand: k-step s = (input component, half of the accumulator registers); lane (column = (out comp, k2), h)
  // holds K = 8 h + i  <->  accumulator register e = 8 (s & 1) + i of a lane with that h: n2 = (e & 3) + 8 (e >> 2) + 4 h
  for (int s = 0; s < 4; ++s)
    for (int ln = 0; ln < 64; ++ln) {
      const int col = ln & 31, hh = ln >> 5, kk2 = col & 15, oc = col >> 4, ic = s >> 1;
      double v[8];
      for (int i = 0; i < 8; ++i) {
        const int e = 8 * (s & 1) + i, n2 = (e & 3) + 8 * (e >> 2) + 4 * hh;
        double c, sn;
        cs((long)n2 * kk2, 32, &c, &sn);
        // (ar + i ai)(c - i sn): re = ar c + ai sn, im = ai c - ar sn
        v[i] = kBdTap * (oc == 0 ? (ic == 0 ? c : sn) : (ic == 0 ? -sn : c));
      }
      bd_planes(v, &ef[((size_t)(2 * s + 0) * 64 + ln) * 4], &ef[((size_t)(2 * s + 1) * 64 + ln) * 4]);
    }
  // twiddles: [t2][ep][lane] = {Re, Im of W2048^(n2 k1) for e = 2 ep, the same for e = 2 ep + 1} x kBdMid
  std::vector<float> tw((size_t)kBdTwF4 * 4);
  for (int t2 = 0; t2 < 2; ++t2)
    for (int ep = 0; ep < 8; ++ep)
      for (int ln = 0; ln < 64; ++ln) {
        const int k1 = 32 * t2 + (ln & 31), hh = ln >> 5;
        for (int d = 0; d < 2; ++d) {
          const int e = 2 * ep + d, n2 = (e & 3) + 8 * (e >> 2) + 4 * hh;
          double c, s;
          cs((long)n2 * k1, 2048, &c, &s);
          tw[(((size_t)(t2 * 8 + ep) * 64 + ln) * 4) + 2 * d] = (float)(kBdMid * c);
          tw[(((size_t)(t2 * 8 + ep) * 64 + ln) * 4) + 2 * d + 1] = (float)(-kBdMid * s);
        }
      }
  // mel: weights x (1/2)^2 (Hann taps applied to 2 Xw) / kBdOut^2; run structure of the two-filters-per-bin bank
  MelPairs mp = mel_pairs();
  if (!mp.ok) { set_error("lipasr_mfcc_plan: mel filter bank is not a two-filters-per-bin bank"); return LIPASR_EUNSUPPORTED; }
  if (mp.wlo[0] != 0.0f || mp.whi[0] != 0.0f || mp.wlo[1024] != 0.0f || mp.whi[1024] != 0.0f) {
    set_error("lipasr_mfcc_plan: mel bank has weight on bin 0 or 1024"); return LIPASR_EUNSUPPORTED;
  }
  std::vector<float> wlo(1024), whi(1024);
  const double wscale = 0.25 / ((double)kBdOut * (double)kBdOut);
  for (int k = 0; k < 1024; ++k) { wlo[k] = (float)(mp.wlo[k] * wscale); whi[k] = (float)(mp.whi[k] * wscale); }
  std::vector<int> run_of(1024, 0);
  for (int m = 0; m < kNMels; ++m)
    for (int k = mp.start[m]; k < mp.start[m] + mp.len[m] && k < 1024; ++k) run_of[k] = m;
  std::vector<unsigned long long> smask(16, 0ull);
  for (int k = 0; k < 1024; ++k)
    if ((k & 15) == 0 || run_of[k] != run_of[k - 1]) smask[k & 15] |= 1ull << (k >> 4);
  auto swz = [](int k) { return (k & ~15) | ((((k >> 2) & 3) ^ ((k >> 6) & 3)) << 2) | (k & 3); };
  std::vector<int> mpos((size_t)kNMels * 8, kBdDummy);
  for (int m = 0; m < kNMels; ++m) {
    for (int part = 0; part < 2; ++part) {  // part 0: run m of the lower-filter plane; part 1: run m - 1 of the upper-filter plane
      const int r = m - part;
      if (r < 0 || mp.len[r] == 0) continue;
      const int s0 = mp.start[r], e0 = std::min(mp.start[r] + mp.len[r] - 1, 1023);
      if (e0 < s0) continue;
      int n = 0;
      int* dst = &mpos[((size_t)m * 2 + part) * 4];
      dst[n++] = swz(e0) + 1024 * part;
      for (int c = (e0 >> 4) - 1; c >= (s0 >> 4); --c) {
        if (n >= 3) { set_error("lipasr_mfcc_plan: a mel run spans more than three 16-bin groups"); return LIPASR_EUNSUPPORTED; }
        dst[n++] = swz(16 * c + 15) + 1024 * part;
      }
    }
  }
  int rc;
  if ((rc = bd_upload(&t->cfrag, cf.data(), cf.size() * 4)) != LIPASR_OK || (rc = bd_upload(&t->efrag, ef.data(), ef.size() * 4)) != LIPASR_OK ||
      (rc = bd_upload(&t->tw, tw.data(), tw.size() * 4)) != LIPASR_OK || (rc = bd_upload(&t->wlo, wlo.data(), wlo.size() * 4)) != LIPASR_OK ||
      (rc = bd_upload(&t->whi, whi.data(), whi.size() * 4)) != LIPASR_OK || (rc = bd_upload(&t->smask, smask.data(), smask.size() * 8)) != LIPASR_OK ||
      (rc = bd_upload(&t->mpos, mpos.data(), mpos.size() * 4)) != LIPASR_OK) {
    bdft_tables_free(t);
    return rc;
  }
  return LIPASR_OK;
}

bool bdft_can_fuse_dct(int n_frames, int seg_frames, int L) {
  seg_frames = std::max(4, (seg_frames + 3) & ~3);
  return seg_frames >= n_frames && n_frames <= 64 && L >= 1 && L <= 64;
}

int launch_stft_bdft(const StftArgs& st, const BdftTables& t, int batch, int seg_frames, const BdftDct* dct, hipStream_t stream) {
  BdftArgs a;
  a.st = st;
  a.L = 0; a.dct_frag = nullptr; a.aff_mean = nullptr; a.aff_scale = nullptr; a.out = nullptr;
  if (dct && bdft_can_fuse_dct(st.n_frames, seg_frames, dct->L)) {
    a.L = dct->L; a.dct_frag = dct->dct_frag; a.aff_mean = dct->aff_mean; a.aff_scale = dct->aff_scale; a.out = dct->out;
  }
  a.cfrag = t.cfrag; a.efrag = t.efrag; a.tw = t.tw; a.wlo = t.wlo; a.whi = t.whi; a.smask = t.smask; a.mpos = t.mpos;
  seg_frames = std::max(4, (seg_frames + 3) & ~3);
  a.seg_frames = seg_frames;
  const size_t lds = (size_t)kBdLdsFloats * sizeof(float);
  static bool attr_set_dev[16] = {};  // per device, as the GEMM launchers do (round 4 set the attribute on every launch: ADVICE r4)
  int attr_dev = 0;
  (void)hipGetDevice(&attr_dev);
  if (!attr_set_dev[attr_dev & 15]) {
    LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stft_bdft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set_dev[attr_dev & 15] = true;
  }
  hipLaunchKernelGGL(stft_bdft_kernel, dim3((st.n_frames + seg_frames - 1) / seg_frames, batch), dim3(256), lds, stream, a);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

}  // namespace lipasr
