// K2: the dense classifier on fp32 MFMA (gfx950), plus BatchNorm / dropout / softmax-CE kernels and
// the plan-level forward / backward / predict / attack sequences.
//
// GEMM design (v_mfma_f32_32x32x2_f32, exact fp32 fma chains):
//   The classifier's GEMMs are small (M = batch 512..1024, N <= 1024, K <= 1024): with one 32x32
//   accumulator per wavefront the time of a tile is (K/2) MFMAs * 64 cycles whatever M and N are,
//   so the lever is K, not the tile.  One workgroup = one 32x32 output tile, its 4 wavefronts split
//   K four ways (16-deep chunks, round-robin), operands go straight from global/L2 to VGPRs in MFMA
//   layout (no LDS staging, no barrier in the main loop, next chunk prefetched behind the MFMAs),
//   the four partial tiles meet in LDS once and 256 threads run the fused epilogue with float4
//   stores.  K order inside a chunk is permuted (lane half h takes k0+8h..k0+8h+7) so that
//   K-contiguous operands load as two float4 per lane; both operands use the same permutation.
//
//   Epilogues fuse: bias (+ReLU), inference BatchNorm affine, the ReLU/BN backward mask of the
//   inference-mode input gradient, and the FGSM/PGD sign step (K4) on the last backward GEMM.
#include "mlp.h"
#include <type_traits>

namespace lipasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum Epi {
  EPI_STORE = 0,
  EPI_BIAS = 1,
  EPI_BIAS_RELU = 2,
  EPI_BIAS_RELU_BN = 3,
  EPI_DZ_INFER = 4,
  EPI_SIGNSTEP = 5,
  EPI_BIAS_RELU_STATS = 6,  // training forward: a = relu(acc + b) and per-tile column sums of a, a^2
  EPI_DH_STATS = 7,         // training backward: g = acc * dropout and per-tile column sums of g, g * xhat
  EPI_DZ_NOBN = 8,          // training backward through Dropout -> ReLU without BatchNorm
  EPI_BIAS_SOFTMAX_CE = 9,  // last layer (N <= 32, one column tile): logits, softmax, CE loss and (p - y) / B in one
  EPI_BIAS_RELU_BNX = 10,   // round 5, training forward: a = relu(acc + b), BatchNorm statistics exchanged between the row tiles of
                            // the column block inside the launch, h = dropout(BN(a)) -- no apply kernel
  EPI_DH_BNX = 11           // round 5, training backward: g = acc * dropout, sums of g, g xhat exchanged, dz = BN/ReLU backward
};

// ---------------------------------------------------------------------------------------------
// dropout multiplier: 0 or 1/(1-rate), Philox keyed by (seed; element/4, layer, step)
// ---------------------------------------------------------------------------------------------
struct DropArgs {
  int mode;  // 0 off, 1 philox, 2 external
  float rate;
  uint64_t seed;
  const int* step_dev;
  int layer;
  const float* mask;
};

__device__ __forceinline__ float dropout_mult(const DropArgs& d, int step, size_t e) {
  if (d.mode == 0 || d.rate <= 0.0f) return 1.0f;
  if (d.mode == 2) return d.mask ? d.mask[e] : 1.0f;
  uint32_t o[4];
  Philox::gen(d.seed, (uint64_t)(e >> 2), (uint32_t)d.layer, (uint32_t)step, o);
  const float u = Philox::u01(o[e & 3]);
  return u > d.rate ? 1.0f / (1.0f - d.rate) : 0.0f;
}


struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc;
  int epi;
  const float* bias;
  const float* gamma;
  const float* beta;
  const float* mmean;
  const float* mvar;
  float* aux;        // EPI_BIAS_RELU_BN: optional post-ReLU store; EPI_DZ_INFER: post-ReLU activations (read)
  const float* x0;   // EPI_SIGNSTEP
  float* x_adv;
  float alpha, eps;
  float* part;             // *_STATS: [2][gridDim.y][N] per-row-tile column partial sums
  const float* save_mean;  // EPI_DH_STATS: batch mean [N], rstd at +N
  DropArgs drop;           // EPI_DH_STATS / EPI_DZ_NOBN
  int ones_row;            // AMODE 1 only: row M-1 of op(A) is all ones (bias gradient = column sums of B)
  float* extra_out;        // its output row goes here instead of C
  // EPI_BIAS_SOFTMAX_CE (what softmax_ce_kernel computes, fused): labels in, the rest optional outputs
  const float* y;          // [M][N] one-hot
  float inv_batch;
  float* prob;             // [M][N]
  float* dz;               // [M][N] (p - y) * inv_batch
  float* loss_rows;        // [M]
  float* correct_rows;     // [M]
  // 0: exact fp32 (v_mfma_f32_32x32x2_f32).  1: operands rounded to bf16 (RNE) at the MFMA, fp32 accumulate
  // (v_mfma_f32_32x32x16_bf16): BASELINE config 2's arithmetic; memory stays fp32.
  int bf16;
  const unsigned* sa_dyn;  // arithmetic mode 2: the operand's largest magnitude (float bits, written by its producer's epilogue): the
  const unsigned* sb_dyn;  // scale is derived from it at run time (gradients: their size is not known beforehand); else sa / sb
  unsigned* amax_out;      // EPI_DH_BNX: max |dz| of this launch is folded into this word (atomic max of float bits)
  unsigned* amax_zero;     // forward launches: workgroup (0, 0) clears this word (the backward pass of the same step fills it)
  float sa, sb;       // arithmetic mode 2: powers of two that bring op(A) and B into fp16's range before the split (the accumulator is divided by sa sb)
  int lds_min_tiles;  // host side only: 64x64 tiles from which launch_gemm takes the LDS-tiled kernel (0 = the default)
  const float* zeros; // >= 16 bytes of zeros in device memory (the ring kernel's source for k >= K in the last k-step), or null
  int cus;            // host side: CUs the launch may use (the plan's budget; 0 = unknown, the whole device)
  int ring;           // host side / grouped launch: this problem takes the LDS-DMA ring tile (mode 2, ring_legal)
  int xcd_map;        // 1: workgroup -> tile by xcd_tile() (a compact patch of the tile grid per XCD); 0: blockIdx as it comes
  // EPI_BIAS_RELU_BNX / EPI_DH_BNX (the exchange epilogue)
  unsigned long long* xc_gran;  // [32-column block][xc_rt_max][128] {tag, value}
  unsigned* xc_ctrl;            // [32-column block][32]: word 0 generation, word 1 arrivals
  int* xc_err;
  int xc_rt_max;
  int Bstat;                    // rows the statistics are taken over
  float grad_scale;             // EPI_DH_BNX: factor on dgamma / dbeta
  float* h_out;                 // EPI_BIAS_RELU_BNX: BatchNorm + dropout output (C receives the post-ReLU activations)
  float* mmean_w;               // EPI_BIAS_RELU_BNX: moving statistics (updated by row tile 0), saved batch mean | rstd
  float* mvar_w;
  float* save_w;
  float* dgamma;                // EPI_DH_BNX
  float* dbeta;
};

// eight consecutive-k fp32 operand values of a lane -> one bf16 fragment (lane (r, h) holds k = 8 h + j, j < 8)
__device__ __forceinline__ bf16x8 to_bf16x8(const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
  return o;
}

// Arithmetic mode 2 (round 5): fp32-accurate products on the fp16 matrix instruction.  Each operand value x (scaled by a power of two
// that keeps it inside fp16's range) is split into two fp16 planes, hi = RNE(x) and lo = RNE(x - hi) (v_fma_mix: the subtraction
// reads hi as fp16), and a product is hi hi + hi lo + lo hi accumulated in fp32: products of fp16 numbers are exact in fp32, what
// is lost is the 2^-22 of the two-plane representation and the lo lo term -- the technique of the resampler and the block-DFT
// STFT (mfcc.hip, stft_bdft.hip), with three v_mfma_f32_32x32x16_f16 of 32 cycles per 16-deep chunk in place of eight
// v_mfma_f32_32x32x2_f32 of 64.  The low plane is ONE asm statement ending in s_nop 1 (a vector-ALU result needs two wait states
// before a matrix instruction reads it, and the hazard recogniser does not look inside inline asm).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
// UNIT: the operand runs unscaled (activations): hi by v_cvt_pk_f16_f32, 12 vector instructions per 8 values.  Otherwise the power-of-two
// scale rides in the conversions themselves, hi = f16(x s + 0) and lo = f16(x s - hi) on v_fma_mix (the scale from an SGPR): 16
// instructions.  (The first version multiplied in front of a run-time `scale != 1` test, which the compiler turned into a multiply
// AND two selects per pair of values: ~26 instructions per split, and the ring kernels are bound by vector-instruction issue.)
template <bool UNIT>
__device__ __forceinline__ void split8(const float (&x)[8], const float scale, f16x8& hi, f16x8& lo) {
  unsigned h[4], l[4];
  if constexpr (UNIT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){x[2 * i], x[2 * i + 1]}, f16x2v));
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
  } else {
    const float s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scale)));  // (wave-uniform by construction)
    asm("v_fma_mixlo_f16 %0, %8, %16, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %10, %16, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %12, %16, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %14, %16, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %9, %16, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %11, %16, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %13, %16, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %15, %16, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %4, %8, %16, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %5, %10, %16, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %6, %12, %16, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %7, %14, %16, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %4, %9, %16, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %5, %11, %16, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %6, %13, %16, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %7, %15, %16, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "s"(s));
  }
  hi = __builtin_bit_cast(f16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(f16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// mode 2, operands whose size is not known beforehand (gradients): the producer's epilogue leaves max |x| as float bits; the
// scale 2^(14 - e) with max = f 2^e, f in [0.5, 1), puts the largest value in [2^13, 2^14) -- a factor 4 under fp16's 65504
// The maximum lives in kAmaxSlots words, one per 64-byte line: 2048 wavefronts folding their maxima into ONE word cost the backward
// kernels 10-22 us each (atomics execute at the memory side, one address serialises them); spread over 64 lines they run side by
// side, and a consumer reads the 64 words with one coalesced... strided load per wavefront and a wave maximum.
constexpr int kAmaxSlots = 64, kAmaxStride = 16;  // words
__device__ __forceinline__ float scale_from_amax(const unsigned* p, const float fallback) {
  if (!p) return fallback;
  const float a = wave_max(__uint_as_float(p[(threadIdx.x & 63) * kAmaxStride]));
  if (!(a > 0.0f) || !(a < INFINITY)) return 1.0f;  // all zero, or NaN / inf (which then propagate as they should)
  int e = 0;
  (void)frexpf(a, &e);
  return ldexpf(1.0f, 14 - e);
}
__device__ __forceinline__ void amax_publish(unsigned* out, float m) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) {
    const unsigned w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    atomicMax(out + (w & (kAmaxSlots - 1)) * kAmaxStride, __float_as_uint(m));  // (non-negative floats order like their bits; NaN is the largest)
  }
}
// workgroup (0, 0) of a forward launch clears the words the backward pass of the same step will fold into
__device__ __forceinline__ void amax_clear(unsigned* out) {
  if (threadIdx.x < kAmaxSlots) out[threadIdx.x * kAmaxStride] = 0u;
}
// one 16-deep chunk: acc += a b on three fp16 matrix instructions
template <bool UA = false>  // UA: the A operand is unscaled (sa == 1: activations)
__device__ __forceinline__ f32x16 mfma_split(const float (&a)[8], const float (&b)[8], const float sa, const float sb, f32x16 acc) {
  f16x8 ah, al, bh, bl;
  split8<UA>(a, sa, ah, al);
  split8<false>(b, sb, bh, bl);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  return acc;
}

// AMODE/BMODE 0: K contiguous in memory (operand(i,k) = P[i*ld + k]); 1: K strided (P[k*ld + i]).
template <int MODE>
__device__ __forceinline__ void load_frag(const float* __restrict__ P, int ld, int idx, int kb, int K, bool vec,
                                          float (&f)[8], bool ones = false) {
  if (MODE == 0) {
    const float* p = P + (size_t)idx * ld + kb;
    if (vec && kb + 8 <= K) {
      const float4 lo = *reinterpret_cast<const float4*>(p);
      const float4 hi = *reinterpret_cast<const float4*>(p + 4);
      f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
      f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = (kb + q < K) ? p[q] : 0.0f;
    }
  } else {
    if (ones) {
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = (kb + q < K) ? 1.0f : 0.0f;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = (kb + q < K) ? P[(size_t)(kb + q) * ld + idx] : 0.0f;
    }
  }
}

// One output element: returns the value to store; s1/s2 receive the column statistics of the *_STATS epilogues.
__device__ __forceinline__ float epilogue_elem(const GemmArgs& g, int step, int gm, int gn, float v, float& s1, float& s2) {
  switch (g.epi) {
    case EPI_BIAS:
      return v + g.bias[gn];
    case EPI_BIAS_RELU:
      return fmaxf(v + g.bias[gn], 0.0f);
    case EPI_BIAS_RELU_STATS: {
      const float a = fmaxf(v + g.bias[gn], 0.0f);
      s1 = a;
      s2 = a * a;
      return a;
    }
    case EPI_BIAS_RELU_BN: {
      const float a = fmaxf(v + g.bias[gn], 0.0f);
      if (g.aux) g.aux[(size_t)gm * g.ldc + gn] = a;
      if (g.gamma) return (a - g.mmean[gn]) / sqrtf(g.mvar[gn] + kBnEps) * g.gamma[gn] + g.beta[gn];
      return a;
    }
    case EPI_DZ_INFER: {
      const float s = g.gamma ? g.gamma[gn] / sqrtf(g.mvar[gn] + kBnEps) : 1.0f;
      return g.aux[(size_t)gm * g.ldc + gn] > 0.0f ? v * s : 0.0f;
    }
    case EPI_DH_STATS: {
      const size_t e = (size_t)gm * g.ldc + gn;
      const float gg = v * dropout_mult(g.drop, step, e);
      const float xh = (g.aux[e] - g.save_mean[gn]) * g.save_mean[g.N + gn];
      s1 = gg;
      s2 = gg * xh;
      return gg;
    }
    case EPI_DZ_NOBN: {
      const size_t e = (size_t)gm * g.ldc + gn;
      return g.aux[e] > 0.0f ? v * dropout_mult(g.drop, step, e) : 0.0f;
    }
    case EPI_SIGNSTEP: {
      const size_t i = (size_t)gm * g.ldc + gn;
      const float sg = (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f);  // NaN -> 0, as ART zeroes NaN gradients
      const float x0 = g.x0[i];
      const float xa = g.x_adv[i] + g.alpha * sg;
      if (isinf(g.eps)) return xa;
      return x0 + fminf(fmaxf(xa - x0, -g.eps), g.eps);
    }
    default:
      return v;
  }
}


// ---------------------------------------------------------------------------------------------
// Round 5: the exchange epilogue.  Training-mode BatchNorm needs column statistics over ALL rows of the batch, i.e. over
// every row tile of a column block; until round 4 the GEMM left per-tile partial sums and a second kernel (bn_apply_*) summed
// them and transformed the tile -- a launch boundary plus a cold round trip for the activations it had just written
// (8.4 / 5.9 us per layer and direction, 72 us of a 352 us step).  Here the row tiles of one column block exchange their
// partial sums inside the launch and every tile finishes its own BatchNorm on the values it still holds in registers:
//   * each tile publishes its 2 x CB partial sums as 8-byte {tag, value} granules (one sc1 store each: the data is the flag),
//   * sweeps the granules of the block's other row tiles until every tag equals this launch's tag (relaxed sc1 loads; the
//     sums are then added in a fixed order in fp64: bitwise reproducible, no float atomics),
//   * and arrives on the block's counter; the last arriver resets it and advances the block's generation, so the next launch
//     (ordered behind this one by the stream) uses the next tag.  Every workgroup reads the generation before it publishes,
//     and the generation cannot move before every workgroup of the block has arrived: all of them use the same tag.
// Needs every workgroup of a column block resident at the same time: the host takes this path only when the whole grid fits
// the CUs the plan's stream may use (bnx_fits), and the sweep is bounded by a wall-clock limit that sets an error word and
// lets the grid drain.  scratch/link_bench.hip (c) prices the exchange alone: 3.7 us (16 row tiles) to 5.5-6.9 us (32).
// ---------------------------------------------------------------------------------------------
struct XcView {
  unsigned long long* gran;
  unsigned* ctrl;
  int* err;
  int rt_max;
};
constexpr long long kXcTimeoutTicks = 200000000LL;  // 2 s of the 100 MHz wall clock
#ifndef LIPASR_XC_POLL_SLEEP
#define LIPASR_XC_POLL_SLEEP 6  // s_sleep units (64 cycles) between two reads of the `published` word by the one polling lane (16 / 6 / 2 measured: config 3 0.3526 / 0.3483 / 0.3490, config 2 0.3109 / 0.3104 / 0.3102)
#endif

// NT threads; CB columns per block (32: the fragment kernel, 64: the LDS-tiled kernel).  mine[2 CB]: this tile's partial sums
// (LDS).  On return tot[2 CB] (LDS, fp64) holds the sums over all n_rt row tiles.  sbuf: LDS, (NT / (2 CB)) x 2 CB doubles.
// this launch's tag for column block bx: the block's generation + 1.  Read at the START of the kernel (the round trip hides behind
// the K loop; any time before this workgroup's own arrival is early enough: the generation cannot move before every row tile
// of the block has arrived)
template <int CB>
__device__ __forceinline__ unsigned xc_tag(const XcView& xc, int bx) {
  return __hip_atomic_load(xc.ctrl + (size_t)bx * (CB / 32) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
}

// `mid`: work of every thread that does not depend on the exchange (the forward pass's store of the post-ReLU activations), run
// while lane 0 waits for the block's other row tiles -- in front of the publish it would sit in the drain the counter waits for
template <int NT, int CB, typename Mid>
__device__ __forceinline__ void xc_exchange(const XcView& xc, int bx, int by, int n_rt, const unsigned want, const float* mine, double* sbuf,
                                            double* tot, Mid mid) {
  constexpr int NI = 2 * CB, PER = NT / NI, MAXK = 64 / PER;
  typedef unsigned long long u64;
  const int tid = threadIdx.x, item = tid % NI, rl = tid / NI;
  const int jblk = bx * (CB / 32);
  unsigned* cw = xc.ctrl + (size_t)jblk * 32;
  u64* g = xc.gran + (size_t)jblk * xc.rt_max * 128;
  if (tid < NI)
    __hip_atomic_store(g + (size_t)by * 128 + tid, ((u64)want << 32) | (u64)__float_as_uint(mine[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // Waiting quietly: the granules carry their own tags, but a workgroup that swept them over and over while the block's other row
  // tiles still computed put 16 KB of sc1 loads on the fabric every 1.5 us -- with 256 tiles on 160 CUs (two rounds) the first
  // round's workgroups polled through the whole second round and the launch took 58 us instead of 36 + 6 (round 5; the guide's
  // polling-cost row).  So a tile counts itself on the block's `published` word once its granule stores have drained, ONE lane
  // polls that word with a pause between reads, and the granules are swept ONCE when it says every row tile is there.
  // (Measured against it: the count without the drain and a sweep that repeats on an old tag -- config 2 0.3409 against 0.3352 ms,
  // config 3 0.404 against 0.400: the repeated sweeps cost more than the drain.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(cw + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  mid();
  if (tid == 0 && n_rt > 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(cw + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n_rt) {
      __builtin_amdgcn_s_sleep(LIPASR_XC_POLL_SLEEP);
      if (wall_clock64() - t0 > kXcTimeoutTicks) { __hip_atomic_store(xc.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
  float v[MAXK];
  {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int t = rl + PER * k;
      v[k] = 0.0f;
      if (t < n_rt) {
        const u64 x = __hip_atomic_load(g + (size_t)t * 128 + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (unsigned)(x >> 32) == want;
        v[k] = __uint_as_float((unsigned)x);
      }
    }
    if (!ok) __hip_atomic_store(xc.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a tag that is not this launch's: never expected
  }
  // arrive now: the returning atomic's round trip runs beside the sums, the normalisation and the stores below
  unsigned old = 0;
  if (tid == 0) old = __hip_atomic_fetch_add(cw + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) s += (double)v[k];  // (slots past n_rt hold 0)
  sbuf[rl * NI + item] = s;
  __syncthreads();
  if (tid < NI) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < PER; ++r) t += sbuf[r * NI + tid];
    tot[tid] = t;
  }
  if (tid == 0 && old == (unsigned)n_rt - 1u) {  // the last row tile of the block: nobody reads the generation or polls any more in this launch
    __hip_atomic_store(cw + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cw + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cw, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}

// BatchNorm element arithmetic shared by the apply kernels and the exchange epilogues
__device__ __forceinline__ void bn_col_stats(double s1, double s2, int Bstat, float& mean, float& var, float& rstd) {
  const double m = s1 / (double)Bstat;
  double v = s2 / (double)Bstat - m * m;
  v = v > 0.0 ? v : 0.0;
  mean = (float)m;
  var = (float)v;
  rstd = (float)(1.0 / sqrt(v + (double)kBnEps));
}

// The tile's R rows x 4 columns per thread after the exchange.  val: a (forward) or g (backward); av: post-ReLU a (backward).
// colp (LDS floats): forward [mean | rstd] per column of the block, backward [dbeta | dgamma].
// The per-column operands of bnx_finish that do not depend on the exchange (forward gamma / beta, backward gamma and the saved mean /
// rstd): requested BEFORE the exchange, so that their round trip runs beside its waits instead of behind them
struct BnxLate {
  float ga[4], b0[4], b1[4];  // forward: gamma, beta, -; backward: gamma, saved mean, saved rstd
};
__device__ __forceinline__ void bnx_late_load(const GemmArgs& g, const int gn, BnxLate& q) {
  const bool fwd = g.epi == EPI_BIAS_RELU_BNX;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool cv = gn + e < g.N;
    q.ga[e] = cv ? g.gamma[gn + e] : 1.0f;
    if (fwd) {
      q.b0[e] = cv ? g.beta[gn + e] : 0.0f;
      q.b1[e] = 0.0f;
    } else {
      q.b0[e] = cv ? g.save_mean[gn + e] : 0.0f;
      q.b1[e] = cv ? g.save_mean[g.N + gn + e] : 1.0f;
    }
  }
}

template <int R>
__device__ __forceinline__ void bnx_finish(const GemmArgs& g, const int step, const int* gm, const int gn, const float (*val)[4],
                                           const float (*av)[4], const float* colp, const int CB, const int c4, const BnxLate& lt) {
  const bool fwd = g.epi == EPI_BIAS_RELU_BNX;
  float ga[4], p0[4], p1[4], be[4], mean[4], rstd[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ga[e] = lt.ga[e];
    p0[e] = colp[c4 + e];
    p1[e] = colp[CB + c4 + e];
    if (fwd) {
      be[e] = lt.b0[e];
      mean[e] = p0[e]; rstd[e] = p1[e];
    } else {
      be[e] = 0.0f;
      mean[e] = lt.b0[e];
      rstd[e] = lt.b1[e];
    }
  }
  const float invB = 1.0f / (float)g.Bstat;
  float omax = 0.0f;  // backward: max |dz| of this thread (arithmetic mode 2 scales the consumers' operand by it)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (gm[r] >= g.M) continue;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t idx = (size_t)gm[r] * g.ldc + gn + e;
      if (fwd) {
        float x = (val[r][e] - mean[e]) * rstd[e] * ga[e] + be[e];
        x *= (gn + e < g.N) ? dropout_mult(g.drop, step, idx) : 0.0f;
        o[e] = x;
      } else {
        const float xh = (av[r][e] - mean[e]) * rstd[e];
        const float d = ga[e] * rstd[e] * (val[r][e] - p0[e] * invB - xh * p1[e] * invB);
        o[e] = av[r][e] > 0.0f ? d : 0.0f;
        if (gn + e < g.N) omax = fmaxf(omax, fabsf(o[e]));
      }
    }
    float* crow = (fwd ? g.h_out : g.C) + (size_t)gm[r] * g.ldc;
    if (gn + 3 < g.N && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(crow) & 15) == 0)) {
      *reinterpret_cast<float4*>(crow + gn) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gn + e < g.N) crow[gn + e] = o[e];
    }
  }
  if (!fwd && g.amax_out) amax_publish(g.amax_out, omax);  // (every thread of the wavefront reaches this point)
}

// one column of the block after the exchange: forward -> [mean | rstd] (+ moving statistics and the saved statistics, by
// row tile 0), backward -> [dbeta | dgamma] (+ the parameter gradients, by row tile 0)
// (mm0, mv0: the column's moving statistics, requested by row tile 0 before the exchange -- bnx_moving_load)
__device__ __forceinline__ void bnx_moving_load(const GemmArgs& g, const int by, const int col, const bool mine, float& mm0, float& mv0) {
  mm0 = 0.0f; mv0 = 0.0f;
  if (mine && g.epi == EPI_BIAS_RELU_BNX && by == 0 && col < g.N) { mm0 = g.mmean_w[col]; mv0 = g.mvar_w[col]; }
}
__device__ __forceinline__ void bnx_column(const GemmArgs& g, const int by, const int col, const int j, const int CB, const double* tot,
                                           float* colp, const float mm0, const float mv0) {
  if (g.epi == EPI_BIAS_RELU_BNX) {
    float mean, var, rstd;
    bn_col_stats(tot[j], tot[CB + j], g.Bstat, mean, var, rstd);
    colp[j] = mean;
    colp[CB + j] = rstd;
    if (by == 0 && col < g.N) {
      g.mmean_w[col] = mm0 * kBnMomentum + mean * (1.0f - kBnMomentum);
      g.mvar_w[col] = mv0 * kBnMomentum + var * (1.0f - kBnMomentum);
      g.save_w[col] = mean;
      g.save_w[g.N + col] = rstd;
    }
  } else {
    const float dbt = (float)tot[j], dg = (float)tot[CB + j];
    colp[j] = dbt;
    colp[CB + j] = dg;
    if (by == 0 && col < g.N) {
      g.dbeta[col] = dbt * g.grad_scale;
      g.dgamma[col] = dg * g.grad_scale;
    }
  }
}

// The per-element operands of the exchange epilogue, requested BEFORE the K-split partial tiles meet in LDS (their round trip
// runs beside that barrier): forward the bias, backward the post-ReLU activation and the column's saved mean / rstd.
struct BnxPre {
  float p0[4], p1[4], p2[4];
};
__device__ __forceinline__ void bnx_prefetch(const GemmArgs& g, const int gm, const int gn, BnxPre& q) {
  const bool rv = gm < g.M;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool cv = rv && gn + e < g.N;
    if (g.epi == EPI_BIAS_RELU_BNX) {
      q.p0[e] = cv ? g.bias[gn + e] : 0.0f;
      q.p1[e] = 0.0f; q.p2[e] = 0.0f;
    } else {
      q.p0[e] = cv ? g.aux[(size_t)gm * g.ldc + gn + e] : 0.0f;
      q.p1[e] = cv ? g.save_mean[gn + e] : 0.0f;
      q.p2[e] = cv ? g.save_mean[g.N + gn + e] : 0.0f;
    }
  }
}

// the element before the exchange: value kept in registers, its two statistics
__device__ __forceinline__ void bnx_elem(const GemmArgs& g, const int step, const bool cv, const int gm, const int gn, const float acc,
                                         const BnxPre& q, const int e, float& val, float& av, float& s1, float& s2) {
  if (g.epi == EPI_BIAS_RELU_BNX) {
    const float a = cv ? fmaxf(acc + q.p0[e], 0.0f) : 0.0f;
    val = a; av = a; s1 = a; s2 = a * a;
  } else {
    const size_t idx = (size_t)gm * g.ldc + gn;
    const float gg = cv ? acc * dropout_mult(g.drop, step, idx) : 0.0f;
    const float a = q.p0[e];
    const float xh = cv ? (a - q.p1[e]) * q.p2[e] : 0.0f;
    val = gg; av = a; s1 = gg; s2 = gg * xh;
  }
}

// XCD-aware workgroup -> tile map (speed only; nothing depends on where a workgroup really runs).  Workgroups are dealt
// round-robin over the 8 XCDs by their linear id, each XCD has its own 4 MiB L2.  With the plain map (bx = id % ntx) XCD x gets the
// column tiles x and x + 8 of EVERY row tile: it reads the whole A operand (3.6-4 MB for the 1024-row layers: its entire L2) and
// an eighth of B.  Here the workgroups of one XCD (ids = x mod 8) take a compact gx x gy patch of the tile grid instead, e.g.
// 8 x 4 tiles of the 16 x 16 grid of layer 1: a quarter of A and half of B, 2.7 MB, so both operands stay in that L2.
// Bijective whenever it applies (ntx divisible by gx, nty by gy); otherwise the plain map.
__device__ __forceinline__ void xcd_tile(const int L, const int ntx, const int nty, int& bx, int& by) {
  // Any grid (round 5; the first version needed ntx, nty divisible by the patch counts, and the 8 x 7 grid of 128-wide weight-gradient
  // tiles fell to "one column of tiles per XCD": seven A panels + one B panel = 4 MB, the whole L2).  The tiles are put in a BLOCKED
  // order -- row groups of height h, inside a group column by column -- and the 8 XCDs take consecutive runs of that order; XCD x runs
  // the workgroups L = x (mod 8) in sequence, so its q-th workgroup takes the q-th tile of its run.  h ~ sqrt(run) makes a run
  // roughly square: about 2 sqrt(run) operand panels instead of run + 1.
  const int total = ntx * nty;
  if (total < 16) { bx = L % ntx; by = L / ntx; return; }
  const int xcd = L & 7, q = L >> 3, base = total >> 3, rem = total & 7;
  const int t = xcd * base + min(xcd, rem) + q;            // position in the blocked order
  int h = (int)(sqrtf((float)(base + (rem ? 1 : 0))) + 0.5f);
  h = max(1, min(h, nty));
  const int per_group = h * ntx, grp = t / per_group, u = t - grp * per_group;
  const int hg = min(h, nty - grp * h);                     // height of this (maybe last, shorter) group
  bx = u / hg;
  by = grp * h + (u - bx * hg);
}

// One workgroup = one 32x32 output tile; its NW wavefronts (4, or 16 for small outputs with a long K) split K
// in 16-deep chunks, round-robin.  Operand fragments go global/L2 -> VGPR directly, one chunk ahead of the MFMAs.
template <int AMODE, int BMODE, int NW, int BF, bool X = false>  // X: the exchange epilogue (its own instances: with it as a run-time
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, const int n_row_tiles) {  // branch every GEMM grew from 50-66 to 83 VGPRs)
  constexpr int TS = 32;
  constexpr int TPR = 8;                       // threads per output row (one float4 each)
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][32][32] + stats [4][8][8]
  float* stat = red + NW * TS * TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = by * TS, n0 = bx * TS;
  const int nch = (g.K + 15) >> 4;
  const bool vecA = (AMODE == 0) && ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const bool vecB = (BMODE == 0) && ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
  const int m_real = g.ones_row ? g.M - 1 : g.M;  // rows of op(A) that exist in memory
  const int row_a = m0 + r;
  const bool aones = (AMODE == 1) && g.ones_row && (row_a == g.M - 1);
  const int ai = min(row_a, m_real - 1);
  const int bj = min(n0 + r, g.N - 1);

  unsigned xtag = 0;
  if constexpr (X) xtag = xc_tag<32>(XcView{g.xc_gran, g.xc_ctrl, g.xc_err, g.xc_rt_max}, bx);
  float rsa = 1.0f, rsb = 1.0f;
  if (BF == 2) { rsa = scale_from_amax(g.sa_dyn, g.sa); rsb = scale_from_amax(g.sb_dyn, g.sb); }
  if (g.amax_zero && bx == 0 && by == 0) amax_clear(g.amax_zero);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  float a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { a0[q] = b0[q] = a1[q] = b1[q] = 0.0f; }
  // one chunk of loads in flight behind the MFMAs (two measured slower: config 2 0.384 -> 0.390 ms, config 5 +4 %)
  int c = wave;
  if (c < nch) {
    load_frag<AMODE>(g.A, g.lda, ai, c * 16 + 8 * h, g.K, vecA, a0, aones);
    load_frag<BMODE>(g.B, g.ldb, bj, c * 16 + 8 * h, g.K, vecB, b0);
  }
  while (c < nch) {
    const int cn = c + NW;
    if (cn < nch) {
      load_frag<AMODE>(g.A, g.lda, ai, cn * 16 + 8 * h, g.K, vecA, a1, aones);
      load_frag<BMODE>(g.B, g.ldb, bj, cn * 16 + 8 * h, g.K, vecB, b1);
    }
    if (BF == 1) {
      // the chunk's 16 k values are exactly one 32x32x16 bf16 fragment per operand (same lane map as the loads)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to_bf16x8(a0), to_bf16x8(b0), acc, 0, 0, 0);
    } else if (BF == 2) {
      acc = mfma_split(a0, b0, rsa, rsb, acc);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b0[q], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { a0[q] = a1[q]; b0[q] = b1[q]; }
    c = cn;
  }
  if (BF == 2) {
    const float un = 1.0f / (rsa * rsb);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] *= un;
  }
  BnxPre xpre;
  if constexpr (X && NW == 4) bnx_prefetch(g, m0 + (tid >> 3), n0 + (tid & 7) * 4, xpre);
  // C/D map of one 32x32 accumulator: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
    red[wave * TS * TS + row * TS + r] = acc[q];
  }
  __syncthreads();

  if constexpr (X && NW == 4) {
    {
      const int step = g.drop.step_dev ? *g.drop.step_dev : 0;
      const int tcol = tid & 7, row = tid >> 3, c4 = tcol * 4, gn = n0 + c4;
      const int gm1[1] = {m0 + row};
      float4 s = *reinterpret_cast<const float4*>(red + row * TS + c4);
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(red + w * TS * TS + row * TS + c4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      const float accv[4] = {s.x, s.y, s.z, s.w};
      float val[1][4], av[1][4], c1[4], c2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bnx_elem(g, step, gm1[0] < g.M && gn + e < g.N, gm1[0], gn + e, accv[e], xpre, e, val[0][e], av[0][e], c1[e], c2[e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = TPR; o < 64; o <<= 1) {
          c1[e] += __shfl_xor(c1[e], o, 64);
          c2[e] += __shfl_xor(c2[e], o, 64);
        }
      }
      if (lane < TPR) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          stat[(wave * TPR + lane) * 8 + e] = c1[e];
          stat[(wave * TPR + lane) * 8 + 4 + e] = c2[e];
        }
      }
      __syncthreads();  // (also: every read of `red` is done, it is carved up below)
      float* mine = red;                                       // [2][32]
      float* colp = red + 64;                                  // [2][32]
      double* sbuf = reinterpret_cast<double*>(red + 128);     // [4][64]
      double* tot = sbuf + 4 * 64;                             // [64]
      if (tid < 2 * TS) {
        const int which = tid / TS, col = tid % TS, l4 = col >> 2, e = col & 3;
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += stat[(w * TPR + l4) * 8 + which * 4 + e];
        mine[tid] = t;
      }
      __syncthreads();
      BnxLate late;
      bnx_late_load(g, gn, late);
      float mm0, mv0;
      bnx_moving_load(g, by, n0 + tid, tid < TS, mm0, mv0);
      XcView xc{g.xc_gran, g.xc_ctrl, g.xc_err, g.xc_rt_max};
      xc_exchange<256, 32>(xc, bx, by, g.Bstat < 0 ? 0 : n_row_tiles, xtag, mine, sbuf, tot, [&]() {  // (Bstat < 0: timing probe, below)
        if (g.epi == EPI_BIAS_RELU_BNX && gm1[0] < g.M) {  // the post-ReLU activations: the backward pass reads them
          float* crow = g.C + (size_t)gm1[0] * g.ldc;
          if (gn + 3 < g.N && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(crow) & 15) == 0)) {
            *reinterpret_cast<float4*>(crow + gn) = make_float4(val[0][0], val[0][1], val[0][2], val[0][3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gn + e < g.N) crow[gn + e] = val[0][e];
          }
        }
      });
      if (tid < TS) bnx_column(g, by, n0 + tid, tid, TS, tot, colp, mm0, mv0);
      __syncthreads();
      bnx_finish<1>(g, step, gm1, gn, val, av, colp, TS, c4, late);
      return;
    }
  }

  const bool stats = (g.epi == EPI_BIAS_RELU_STATS) || (g.epi == EPI_DH_STATS);
  float cs1[4] = {0.f, 0.f, 0.f, 0.f}, cs2[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < 256) {
    const int step = g.drop.step_dev ? *g.drop.step_dev : 0;
    const int tcol = tid & 7, row = tid >> 3;
    const int c4 = tcol * 4;
    const int gn = n0 + c4;
    float4 s = *reinterpret_cast<const float4*>(red + row * TS + c4);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + w * TS * TS + row * TS + c4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int gm = m0 + row;
    if (g.epi == EPI_BIAS_SOFTMAX_CE) {
      // the row's (<= 32) logits sit in the 8 consecutive lanes that share `row`: butterfly over lane bits 0..2.
      // Same definitions as softmax_ce_kernel (first maximum wins ties); every lane takes part in the shuffles.
      const bool rv = gm < g.M;
      float z[4];
      float mx = -INFINITY;
      int am = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool cv = rv && (gn + e < g.N);
        z[e] = cv ? (&s.x)[e] + g.bias[gn + e] : -INFINITY;
        if (z[e] > mx) { mx = z[e]; am = gn + e; }
      }
#pragma unroll
      for (int o2 = 1; o2 < 8; o2 <<= 1) {
        const float omx = __shfl_xor(mx, o2, 64);
        const int oam = __shfl_xor(am, o2, 64);
        if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
      }
      float se = 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) se += (z[e] > -INFINITY) ? expf(z[e] - mx) : 0.0f;
#pragma unroll
      for (int o2 = 1; o2 < 8; o2 <<= 1) se += __shfl_xor(se, o2, 64);
      const float lse = logf(se), inv = 1.0f / se;
      float loss = 0.0f, ymax = -INFINITY;
      int ay = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (rv && gn + e < g.N) {
          const size_t idx = (size_t)gm * g.N + gn + e;
          const float zs = z[e] - mx;
          const float pc = expf(zs) * inv;
          g.C[(size_t)gm * g.ldc + gn + e] = z[e];
          if (g.prob) g.prob[idx] = pc;
          if (g.y) {
            const float yc = g.y[idx];
            if (yc != 0.0f) loss -= yc * (zs - lse);
            if (yc > ymax) { ymax = yc; ay = gn + e; }
            if (g.dz) g.dz[idx] = (pc - yc) * g.inv_batch;
          }
        }
      }
#pragma unroll
      for (int o2 = 1; o2 < 8; o2 <<= 1) {
        loss += __shfl_xor(loss, o2, 64);
        const float oym = __shfl_xor(ymax, o2, 64);
        const int oay = __shfl_xor(ay, o2, 64);
        if (oym > ymax || (oym == ymax && oay < ay)) { ymax = oym; ay = oay; }
      }
      if (rv && tcol == 0) {
        if (g.loss_rows) g.loss_rows[gm] = loss;
        if (g.correct_rows) g.correct_rows[gm] = (am == ay) ? 1.0f : 0.0f;
      }
    } else if (gm < g.M) {
      const float v[4] = {s.x, s.y, s.z, s.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t1 = 0.0f, t2 = 0.0f;
        o[e] = (gn + e < g.N) ? epilogue_elem(g, step, gm, gn + e, v[e], t1, t2) : 0.0f;
        cs1[e] = t1;
        cs2[e] = t2;
      }
      float* crow;
      if (g.ones_row && gm == g.M - 1) crow = g.extra_out;
      else crow = (g.epi == EPI_SIGNSTEP ? g.x_adv : g.C) + (size_t)gm * g.ldc;
      if (gn + 3 < g.N && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(crow) & 15) == 0)) {
        *reinterpret_cast<float4*>(crow + gn) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gn + e < g.N) crow[gn + e] = o[e];
      }
    }
  }
  if (stats) {
    // column sums over the tile's 32 rows: lanes with equal tcol inside a wavefront (8 rows), then 4 wavefronts
    if (tid < 256) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = TPR; o < 64; o <<= 1) {
          cs1[e] += __shfl_xor(cs1[e], o, 64);
          cs2[e] += __shfl_xor(cs2[e], o, 64);
        }
      }
      if (lane < TPR) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          stat[(wave * TPR + lane) * 8 + e] = cs1[e];
          stat[(wave * TPR + lane) * 8 + 4 + e] = cs2[e];
        }
      }
    }
    __syncthreads();
    if (tid < 2 * TS) {
      const int which = tid / TS, col = tid % TS;
      const int l4 = col >> 2, e = col & 3;
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += stat[(w * TPR + l4) * 8 + which * 4 + e];
      if (n0 + col < g.N) g.part[((size_t)which * n_row_tiles + by) * g.N + n0 + col] = t;
    }
  }
}

template <int AMODE, int BMODE, int NW, int BF = 0, bool X = false>  // BF: operands rounded to bf16 at the MFMA (compile-time: a
__global__ __launch_bounds__(64 * NW) void gemm_f32_kernel(GemmArgs g) {  // run-time switch cost the fp32 path 6 %)
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_map) xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, bx, by);
  gemm_tile<AMODE, BMODE, NW, BF, X>(g, bx, by, gridDim.y);
}

// Several independent GEMMs of one (AMODE, BMODE) in ONE launch: the six weight-gradient GEMMs of a training step
// (outputs from 880x1024 down to 64x10, all with K = batch) fill the chip together instead of running as six
// mostly latency-bound launches.  Block b belongs to the problem whose tile range contains it.
constexpr int kMaxGroup = 8;
struct GemmGroup {
  int n;
  int tile_start[kMaxGroup + 1];
  GemmArgs g[kMaxGroup];
};

template <int AMODE, int BMODE, int NW, int BF = 0>
__global__ __launch_bounds__(64 * NW) void gemm_f32_grouped_kernel(GemmGroup grp) {
  int p = 0;
  while (p + 1 < grp.n && (int)blockIdx.x >= grp.tile_start[p + 1]) ++p;
  const GemmArgs& g = grp.g[p];
  const int local = blockIdx.x - grp.tile_start[p];
  const int ntx = (g.N + 31) / 32, nty = (g.M + 31) / 32;
  int bx = local % ntx, by = local / ntx;
  if (g.xcd_map && (grp.tile_start[p] & 7) == 0) xcd_tile(local, ntx, nty, bx, by);
  gemm_tile<AMODE, BMODE, NW, BF>(g, bx, by, nty);
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled variant for the large GEMMs: one workgroup (512 threads, 8 wavefronts) = one 64x64 tile.  K advances
// in 32-deep stages through a double-buffered LDS image stored k-major ([k][m] and [k][n], row stride 68
// floats): MFMA operand reads are unit-stride ds_read_b32 (conflict-free) and every operand element is
// fetched from L2 once per workgroup instead of once per 32x32 tile.  Wavefronts 0-3 take k 0..15 of each
// stage for the four 32x32 quadrants, wavefronts 4-7 take k 16..31: two wavefronts per SIMD, so one's LDS
// latency hides behind the other's MFMAs.  Global loads for stage t+1 are issued before the MFMAs of stage t
// and written to the other LDS buffer afterwards: one barrier per stage.  The two K halves meet in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int kLdsBK = 32, kLdsLD = 68;

template <int MODE>  // 0: operand(i,k) = P[i*ld + k] (K contiguous), 1: P[k*ld + i]
__device__ __forceinline__ float4 tile_fetch(const float* __restrict__ P, int ld, int i0, int i_real, int k0, int K,
                                             bool ones_last, int i_last, int tid) {
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  const bool al = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(P) & 15) == 0);
  if (MODE == 0) {
    const int i = i0 + (tid >> 3);                     // 64 rows, 8 float4 (32 k) per row
    const int kq = k0 + (tid & 7) * 4;
    const float* p = P + (size_t)min(i, i_real - 1) * ld + kq;
    if (kq + 3 < K && al) {
      const float4 x = *reinterpret_cast<const float4*>(p);
      t[0] = x.x; t[1] = x.y; t[2] = x.z; t[3] = x.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) t[c] = (kq + c < K) ? p[c] : 0.0f;
    }
  } else {
    const int k = k0 + (tid >> 4);                     // 32 k rows, 16 float4 (64 i) per row
    const int iq = i0 + (tid & 15) * 4;
    if (k < K) {
      const float* p = P + (size_t)k * ld;
      if (iq + 3 < i_real && al) {
        const float4 x = *reinterpret_cast<const float4*>(p + iq);
        t[0] = x.x; t[1] = x.y; t[2] = x.z; t[3] = x.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = iq + c;
          t[c] = (ones_last && i == i_last) ? 1.0f : p[min(i, i_real - 1)];
        }
      }
    }
  }
  return make_float4(t[0], t[1], t[2], t[3]);
}

template <int MODE>
__device__ __forceinline__ void tile_store(float* __restrict__ S, int tid, const float4 v) {
  if (MODE == 0) {
    const int i = tid >> 3, kq = (tid & 7) * 4;
    S[(kq + 0) * kLdsLD + i] = v.x;
    S[(kq + 1) * kLdsLD + i] = v.y;
    S[(kq + 2) * kLdsLD + i] = v.z;
    S[(kq + 3) * kLdsLD + i] = v.w;
  } else {
    const int k = tid >> 4, iq = (tid & 15) * 4;
    *reinterpret_cast<float4*>(S + k * kLdsLD + iq) = v;
  }
}

constexpr int kLdsBKMax = 32;  // 64-row tiles measured slower (30.4 vs 26.3 us on the 1024x1024x880 GEMM)
constexpr size_t lds_gemm_bytes(int bk) {
  return ((size_t)(2 * 2 * bk * kLdsLD > 2 * 64 * 64 ? 2 * 2 * bk * kLdsLD : 2 * 64 * 64) + 8 * 16 * 8) * sizeof(float);
}

// The epilogue of a 64 x 64 tile held as eight 32 x 32 accumulators (wavefront = (K half, quadrant)): the two K halves meet in LDS
// (`red`: the first 32 KB of the workgroup's dynamic LDS, free by now), then bias / ReLU / statistics / the exchange epilogue /
// stores.  Shared by the LDS-tiled kernel and the LDS-DMA ring kernel.
template <bool X>
__device__ __forceinline__ void lds_tile_epilogue(const GemmArgs& g, const f32x16& acc, float* lds, float* stat, const int bx, const int by,
                                                  const int n_row_tiles, const unsigned xtag) {
  constexpr int TS = 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int kh = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int m0 = by * TS, n0 = bx * TS;
  BnxPre xpre[2];
  if constexpr (X) {
    bnx_prefetch(g, m0 + (tid >> 4), n0 + (tid & 15) * 4, xpre[0]);
    bnx_prefetch(g, m0 + (tid >> 4) + 32, n0 + (tid & 15) * 4, xpre[1]);
  }
  // accumulators -> LDS as two 64x64 partial tiles (one per K half), then the shared epilogue shape
  float* red = lds;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = 32 * wi + (q & 3) + 8 * (q >> 2) + 4 * h;
    red[kh * TS * TS + row * TS + 32 * wj + r] = acc[q];
  }
  __syncthreads();
  const bool stats = (g.epi == EPI_BIAS_RELU_STATS) || (g.epi == EPI_DH_STATS);
  const int step = g.drop.step_dev ? *g.drop.step_dev : 0;
  const int tcol = tid & 15, trow = tid >> 4;
  const int c4 = tcol * 4, gn = n0 + c4;
  if constexpr (X) {
    const int gm2[2] = {m0 + trow, m0 + trow + 32};
    float val[2][4], av[2][4], c1[4] = {0.f, 0.f, 0.f, 0.f}, c2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = trow + 32 * pass;
      float4 s = *reinterpret_cast<const float4*>(red + row * TS + c4);
      const float4 s2 = *reinterpret_cast<const float4*>(red + TS * TS + row * TS + c4);
      const float accv[4] = {s.x + s2.x, s.y + s2.y, s.z + s2.z, s.w + s2.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t1, t2;
        bnx_elem(g, step, gm2[pass] < g.M && gn + e < g.N, gm2[pass], gn + e, accv[e], xpre[pass], e, val[pass][e], av[pass][e], t1, t2);
        c1[e] += t1;
        c2[e] += t2;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      c1[e] += __shfl_xor(c1[e], 16, 64); c1[e] += __shfl_xor(c1[e], 32, 64);
      c2[e] += __shfl_xor(c2[e], 16, 64); c2[e] += __shfl_xor(c2[e], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        stat[(wave * 16 + lane) * 8 + e] = c1[e];
        stat[(wave * 16 + lane) * 8 + 4 + e] = c2[e];
      }
    }
    __syncthreads();  // (every read of `red` is done: it is carved up below)
    float* mine = red;                                       // [2][64]
    float* colp = red + 128;                                 // [2][64]
    double* sbuf = reinterpret_cast<double*>(red + 256);     // [4][128]
    double* tot = sbuf + 4 * 128;                            // [128]
    if (tid < 2 * TS) {
      const int which = tid / TS, col = tid % TS, l4 = col >> 2, e = col & 3;
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += stat[(w * 16 + l4) * 8 + which * 4 + e];
      mine[tid] = t;
    }
    __syncthreads();
    BnxLate late;
    bnx_late_load(g, gn, late);
    float mm0, mv0;
    bnx_moving_load(g, by, n0 + tid, tid < TS, mm0, mv0);
    XcView xc{g.xc_gran, g.xc_ctrl, g.xc_err, g.xc_rt_max};
    xc_exchange<512, 64>(xc, bx, by, g.Bstat < 0 ? 0 : n_row_tiles, xtag, mine, sbuf, tot, [&]() {
      if (g.epi != EPI_BIAS_RELU_BNX) return;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if (gm2[pass] >= g.M) continue;
        float* crow = g.C + (size_t)gm2[pass] * g.ldc;
        if (gn + 3 < g.N && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(crow) & 15) == 0)) {
          *reinterpret_cast<float4*>(crow + gn) = make_float4(val[pass][0], val[pass][1], val[pass][2], val[pass][3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gn + e < g.N) crow[gn + e] = val[pass][e];
        }
      }
    });
    if (tid < TS) bnx_column(g, by, n0 + tid, tid, TS, tot, colp, mm0, mv0);
    __syncthreads();
    bnx_finish<2>(g, step, gm2, gn, val, av, colp, TS, c4, late);
    return;
  }
  float cs1[4] = {0.f, 0.f, 0.f, 0.f}, cs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = trow + 32 * pass;
    float4 s = *reinterpret_cast<const float4*>(red + row * TS + c4);
    const float4 s2 = *reinterpret_cast<const float4*>(red + TS * TS + row * TS + c4);
    s.x += s2.x; s.y += s2.y; s.z += s2.z; s.w += s2.w;
    const int gm = m0 + row;
    if (gm < g.M) {
      const float v[4] = {s.x, s.y, s.z, s.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t1 = 0.0f, t2 = 0.0f;
        o[e] = (gn + e < g.N) ? epilogue_elem(g, step, gm, gn + e, v[e], t1, t2) : 0.0f;
        cs1[e] += t1;
        cs2[e] += t2;
      }
      float* crow;
      if (g.ones_row && gm == g.M - 1) crow = g.extra_out;
      else crow = (g.epi == EPI_SIGNSTEP ? g.x_adv : g.C) + (size_t)gm * g.ldc;
      if (gn + 3 < g.N && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(crow) & 15) == 0)) {
        *reinterpret_cast<float4*>(crow + gn) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gn + e < g.N) crow[gn + e] = o[e];
      }
    }
  }
  if (stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cs1[e] += __shfl_xor(cs1[e], 16, 64); cs1[e] += __shfl_xor(cs1[e], 32, 64);
      cs2[e] += __shfl_xor(cs2[e], 16, 64); cs2[e] += __shfl_xor(cs2[e], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        stat[(wave * 16 + lane) * 8 + e] = cs1[e];
        stat[(wave * 16 + lane) * 8 + 4 + e] = cs2[e];
      }
    }
    __syncthreads();
    if (tid < 2 * TS) {
      const int which = tid / TS, col = tid % TS;
      const int l4 = col >> 2, e = col & 3;
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += stat[(w * 16 + l4) * 8 + which * 4 + e];
      if (n0 + col < g.N) g.part[((size_t)which * n_row_tiles + by) * g.N + n0 + col] = t;
    }
  }
}


template <int AMODE, int BMODE, int BF, int BK, bool X = false>  // BK: k rows per LDS tile (32 or 64); X: the exchange epilogue
__device__ __forceinline__ void gemm_lds_tile(const GemmArgs& g, const int bx, const int by, const int n_row_tiles) {
  constexpr int TS = 64;
  constexpr int NF = BK / kLdsBK;  // 32-row fetches per operand and tile
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [buf][A|B][BK][68]; reused as [2][64][64]; then stat
  float* stat = lds + (2 * 2 * BK * kLdsLD > 2 * 64 * 64 ? 2 * 2 * BK * kLdsLD : 2 * 64 * 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int kh = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int m0 = by * TS, n0 = bx * TS;
  const int m_real = g.ones_row ? g.M - 1 : g.M;
  const int nst = (g.K + BK - 1) / BK;
  const bool ones = g.ones_row != 0;
  unsigned xtag = 0;
  if constexpr (X) xtag = xc_tag<64>(XcView{g.xc_gran, g.xc_ctrl, g.xc_err, g.xc_rt_max}, bx);
  float rsa = 1.0f, rsb = 1.0f;
  if (BF == 2) { rsa = scale_from_amax(g.sa_dyn, g.sa); rsb = scale_from_amax(g.sb_dyn, g.sb); }
  if (g.amax_zero && bx == 0 && by == 0) amax_clear(g.amax_zero);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  // Operand tiles travel global/L2 -> registers -> LDS.  A k-step's MFMAs take 0.25 us per 32 k rows, an L2 round trip
  // more: with the fetch of tile t+1 issued at the top of step t and stored at its bottom, every step waited for memory
  // (28 steps x ~0.8 us on the 880-deep layer-1 GEMMs).  The ring below keeps TWO tiles in flight (tile t+2 is
  // requested at the top of step t and stored at the bottom of step t+1): 29.1 -> 26.3 us on that GEMM.
  struct Tile { float4 a[NF], b[NF]; };
  auto fetch = [&](Tile& tl, const int t) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      tl.a[f] = tile_fetch<AMODE>(g.A, g.lda, m0, m_real, t * BK + f * kLdsBK, g.K, ones, g.M - 1, tid);
      tl.b[f] = tile_fetch<BMODE>(g.B, g.ldb, n0, g.N, t * BK + f * kLdsBK, g.K, false, 0, tid);
    }
  };
  auto park = [&](const Tile& tl, const int t) {
    float* An = lds + (t & 1) * 2 * BK * kLdsLD;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      tile_store<AMODE>(An + f * kLdsBK * kLdsLD, tid, tl.a[f]);
      tile_store<BMODE>(An + BK * kLdsLD + f * kLdsBK * kLdsLD, tid, tl.b[f]);
    }
  };
  Tile t0, t1;
  fetch(t0, 0);
  if (nst > 1) fetch(t1, 1);
  park(t0, 0);
  __syncthreads();
  // one k-step: request tile t+2 into the free register slot, multiply tile t out of LDS, park tile t+1 (requested one
  // step ago) in the other LDS buffer
  auto kstep = [&](const int t, Tile& free_slot, const Tile& ready) {
    if (t + 2 < nst) fetch(free_slot, t + 2);
    // this wavefront's K half of the tile: BK/2 rows from (BK/2) kh, in groups of 16.  fp32: lane half h takes
    // k = h + 2 s of a group (one 32x32x2 per s); bf16: k = 8 h + s (one 32x32x16 per group)
    constexpr int kstr = BF ? kLdsLD : 2 * kLdsLD;  // (modes 1 and 2 share the 32x32x16 lane map)
    const int koff = BF ? 8 * h : h;
    const float* base = lds + (t & 1) * 2 * BK * kLdsLD + ((BK / 2) * kh + koff) * kLdsLD + r;
#pragma unroll
    for (int q = 0; q < BK / 32; ++q) {
      const float* As = base + 16 * q * kLdsLD + 32 * wi;
      const float* Bs = base + BK * kLdsLD + 16 * q * kLdsLD + 32 * wj;
      float av[8], bv[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        av[s] = As[s * kstr];
        bv[s] = Bs[s * kstr];
      }
      if (BF == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(to_bf16x8(av), to_bf16x8(bv), acc, 0, 0, 0);
      } else if (BF == 2) {
        acc = mfma_split(av, bv, rsa, rsb, acc);
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
      }
    }
    if (t + 1 < nst) park(ready, t + 1);
    __syncthreads();
  };
  for (int t = 0; t < nst; t += 2) {
    kstep(t, t0, t1);                    // even step: slot 0 is free (tile t is in LDS), slot 1 holds tile t+1
    if (t + 1 < nst) kstep(t + 1, t1, t0);  // odd step: the roles swap
  }
  if (BF == 2) {
    const float un = 1.0f / (rsa * rsb);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] *= un;
  }
  lds_tile_epilogue<X>(g, acc, lds, stat, bx, by, n_row_tiles, xtag);
}


// ---------------------------------------------------------------------------------------------
// The LDS-DMA ring kernel (round 5; arithmetic mode 2 only).  Same 64 x 64 tile, same eight wavefronts (K half, quadrant) and the
// same epilogue as gemm_lds_tile -- what changes is how the operand tiles reach LDS.  With the products on the fp16 matrix
// instruction a 32-deep k-step is ~200 cycles of arithmetic per wavefront, and the register-staged pipeline (global -> VGPR ->
// ds_write, two tiles in flight, 8 + 8 ds_read_b32 per operand group) was bound by the memory round trip per step: the grouped
// weight-gradient launch took 77 us on 128 CUs for ~10 us of arithmetic.  Here every wavefront issues two
// `global_load_lds_dwordx4` per k-step (1 KB each, straight into the ring slot: no staging registers, no LDS store instructions),
// FOUR k-steps live in the 64 KB ring and three are in flight behind the one being multiplied; one workgroup barrier per k-step.
//   * operand stored k-major in memory (P[k ld + i]: both operands of the weight-gradient GEMMs, the kernels in the forward
//     pass): the slot holds [32 k][64 i], one DMA instruction = 4 k rows; fragment reads are unit-stride ds_read_b32.
//   * operand stored K-contiguous (P[i ld + k]: activations, dz, the kernels in the dX GEMMs): the slot holds [64 i][32 k] with the
//     eight 16-byte chunks of a row XOR-swizzled by (i >> 1) & 7 -- an LDS-DMA instruction writes lane l's 16 bytes at l x 16, so
//     the swizzle is applied to the ADDRESS each lane fetches from; a fragment (8 consecutive k of row i) is two ds_read_b128,
//     conflict-free in the hardware's 16-lane groups (MI355X_MICROARCH.md, LDS).
// The DMA instructions are inline asm: the compiler waits for vmcnt(0) in front of every LDS read that follows a
// __builtin_amdgcn_global_load_lds (it cannot tell the slots apart), which would take the three k-steps in flight back to none;
// the waits are counted by hand (two DMA instructions per wavefront and k-step, completed in order).
// Legal when K is a multiple of 32, both leading dimensions are multiples of 4, both bases 16-byte aligned and a k-major
// operand's extent is a multiple of 4 (ring_legal); anything else takes gemm_lds_tile / gemm_tile, which handle every shape.
// ---------------------------------------------------------------------------------------------
#ifndef LIPASR_RING_STAGES
#define LIPASR_RING_STAGES 4
#endif
constexpr int kRingStages = LIPASR_RING_STAGES;
constexpr int kRingTile = 64 * 32;  // floats of one operand tile of one k-step (8 KB)
constexpr size_t ring_gemm_bytes() { return (size_t)(kRingStages * 2 * kRingTile + 8 * 16 * 8) * sizeof(float); }

__device__ __forceinline__ void dma16(const float* gsrc, const unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_byte_addr)
               : "memory");
}

// this lane's source address of operand tile rows/columns i0 .. i0 + 63 at k = 0 (advance by ring_step per k-step)
template <int MODE>
__device__ __forceinline__ const float* ring_src(const float* P, const int ld, const int i0, const int i_real, const int wave, const int lane) {
  if (MODE == 1) {
    const int k = 4 * wave + (lane >> 4), i = min(i0 + (lane & 15) * 4, i_real - 4);
    return P + (size_t)k * ld + i;
  }
  const int il = 8 * wave + (lane >> 3), c = (lane & 7) ^ ((il >> 1) & 7);
  return P + (size_t)min(i0 + il, i_real - 1) * ld + 4 * c;
}
template <int MODE>
__device__ __forceinline__ size_t ring_step(const int ld) { return MODE == 1 ? (size_t)32 * ld : (size_t)32; }

// the 8 consecutive k (16 kh + 8 hh ..) of row / column `il` of an operand tile in a ring slot
template <int MODE>
__device__ __forceinline__ void ring_frag(const float* __restrict__ T, const int il, const int kh, const int hh, float (&v)[8]) {
  if (MODE == 1) {
    const float* q = T + (16 * kh + 8 * hh) * 64 + il;
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = q[s * 64];
  } else {
    const int sw = (il >> 1) & 7, c0 = 4 * kh + 2 * hh;
    const float4 lo = *reinterpret_cast<const float4*>(T + (il * 8 + (c0 ^ sw)) * 4);
    const float4 hi = *reinterpret_cast<const float4*>(T + (il * 8 + ((c0 + 1) ^ sw)) * 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
    v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
  }
}

template <int AMODE, int BMODE, bool X = false, int NL = 0>  // NL: loader wavefronts beside the eight that multiply (0: every wavefront brings its own two pieces)
__device__ __forceinline__ void gemm_ring_tile(const GemmArgs& g, const int bx, const int by, const int n_row_tiles) {
  constexpr int TS = 64, S = kRingStages;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][A | B][2048]; the epilogue reuses the first 32 KB; then stat
  float* stat = lds + S * 2 * kRingTile;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int kh = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int m0 = by * TS, n0 = bx * TS;
  const int m_real = g.ones_row ? g.M - 1 : g.M;
  const int nst = (g.K + 31) >> 5;
  unsigned xtag = 0;
  if constexpr (X) xtag = xc_tag<64>(XcView{g.xc_gran, g.xc_ctrl, g.xc_err, g.xc_rt_max}, bx);
  // K that is no multiple of 32 (the 880 features of layer 1): in the last k-step the lanes whose 16 bytes lie at k >= K fetch zeros
  const int koff_a = AMODE == 1 ? 4 * wave + (lane >> 4) : 4 * ((lane & 7) ^ (((8 * wave + (lane >> 3)) >> 1) & 7));
  const int koff_b = BMODE == 1 ? 4 * wave + (lane >> 4) : 4 * ((lane & 7) ^ (((8 * wave + (lane >> 3)) >> 1) & 7));
  const bool k_tail = (g.K & 31) != 0;
  const float rsa = scale_from_amax(g.sa_dyn, g.sa), rsb = scale_from_amax(g.sb_dyn, g.sb);
  if (g.amax_zero && bx == 0 && by == 0) amax_clear(g.amax_zero);
  const float* pa = ring_src<AMODE>(g.A, g.lda, m0, m_real, wave, lane);
  const float* pb = ring_src<BMODE>(g.B, g.ldb, n0, g.N, wave, lane);
  const size_t sa_step = ring_step<AMODE>(g.lda), sb_step = ring_step<BMODE>(g.ldb);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds) + (unsigned)wave * 1024u;  // this wavefront's 1 KB of an A tile
  auto issue = [&](const int t) {
    const unsigned slot = lds0 + (unsigned)(t % S) * (2u * kRingTile * 4u);
    const bool last = k_tail && t == nst - 1;
    dma16((last && 32 * t + koff_a >= g.K) ? g.zeros : pa, slot);
    dma16((last && 32 * t + koff_b >= g.K) ? g.zeros : pb, slot + kRingTile * 4u);
    pa += sa_step;
    pb += sb_step;
  };
  const int il_ones = (AMODE == 1 && g.ones_row && g.M - 1 >= m0 && g.M - 1 < m0 + TS) ? g.M - 1 - m0 : -1;
  if constexpr (NL > 0) {
    // NL extra wavefronts bring the 16 pieces of a k-step (see the weight-gradient tile: a wavefront that issues LDS-DMA sits in the address
    // path meanwhile); loader L takes the pieces of wavefronts L, L + NL, ... of both operands
    if (wave >= 8) {
      const int L = wave - 8;
      constexpr int NP = 8 / NL;
      const float* sa[NP];
      const float* sb[NP];
      int ka[NP], kb[NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int w = L + NL * q;
        sa[q] = ring_src<AMODE>(g.A, g.lda, m0, m_real, w, lane);
        sb[q] = ring_src<BMODE>(g.B, g.ldb, n0, g.N, w, lane);
        ka[q] = AMODE == 1 ? 4 * w + (lane >> 4) : 4 * ((lane & 7) ^ (((8 * w + (lane >> 3)) >> 1) & 7));
        kb[q] = BMODE == 1 ? 4 * w + (lane >> 4) : 4 * ((lane & 7) ^ (((8 * w + (lane >> 3)) >> 1) & 7));
      }
      const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
      auto issue_l = [&](const int t) {
        const unsigned slot = base + (unsigned)(t % S) * (2u * kRingTile * 4u);
        const bool last = k_tail && t == nst - 1;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const unsigned d = slot + (unsigned)(L + NL * q) * 1024u;
          dma16((last && 32 * t + ka[q] >= g.K) ? g.zeros : sa[q], d);
          dma16((last && 32 * t + kb[q] >= g.K) ? g.zeros : sb[q], d + kRingTile * 4u);
          sa[q] += sa_step;
          sb[q] += sb_step;
        }
      };
      for (int t = 0; t < min(S - 1, nst); ++t) issue_l(t);
      for (int t = 0; t < nst; ++t) {
        const int ahead = min(t + S - 2, nst - 1) - t;  // k-steps requested beyond t: 2 NP instructions each, completed in order
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NP) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (il_ones >= 0 && lane < 4 * NP) lds[(t % S) * 2 * kRingTile + (4 * (L + NL * (lane >> 2)) + (lane & 3)) * 64 + il_ones] = 1.0f;
        __syncthreads();
        if (t + S - 1 < nst) issue_l(t + S - 1);
      }
      __syncthreads();  // (the barrier in front of the epilogue)
      return;
    }
  }
  const int pre = NL > 0 ? 0 : min(S - 1, nst);
  for (int t = 0; t < pre; ++t) issue(t);
  // (the all-ones row of op(A) -- bias gradient of the weight-gradient GEMMs -- does not exist in memory: it is written into the slot)
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  auto k_loop = [&](auto unit_a) {  // (two copies of the loop: an unscaled A operand -- the activations -- splits in 12 instructions instead of 16)
    constexpr bool UA = decltype(unit_a)::value;
    for (int t = 0; t < nst; ++t) {
      float* At = lds + (t % S) * 2 * kRingTile;
      if constexpr (NL == 0) {
        const int ahead = min(t + S - 2, nst - 1) - t;  // k-steps requested beyond t: two DMA instructions each, completed in order
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (il_ones >= 0 && lane < 4) At[(4 * wave + lane) * 64 + il_ones] = 1.0f;  // (this wavefront's own four k rows: they have landed)
      }
      __syncthreads();  // every wavefront's part of k-step t is in LDS, and everybody is done with the slot of k-step t - 1
#if !defined(LIPASR_RING_PROBE) || LIPASR_RING_PROBE != 2   // (timing probes, never shipped: 1 = no arithmetic, 2 = no operand traffic after the prologue)
      if constexpr (NL == 0) {
        if (t + S - 1 < nst) issue(t + S - 1);
      }
#endif
#if !defined(LIPASR_RING_PROBE) || LIPASR_RING_PROBE != 1
      float av[8], bv[8];
      ring_frag<AMODE>(At, 32 * wi + r, kh, hh, av);
      ring_frag<BMODE>(At + kRingTile, 32 * wj + r, kh, hh, bv);
      acc = mfma_split<UA>(av, bv, rsa, rsb, acc);
#endif
    }
  };
  if (rsa == 1.0f) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  {
    const float un = 1.0f / (rsa * rsb);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] *= un;
  }
  __syncthreads();  // the last k-step's fragments are read: the ring becomes the epilogue's `red`
  lds_tile_epilogue<X>(g, acc, lds, stat, bx, by, n_row_tiles, xtag);
}

#ifndef LIPASR_RING_LOADERS
#define LIPASR_RING_LOADERS 0
#endif
constexpr int kRingLoaders = LIPASR_RING_LOADERS;  // 0, 1, 2 or 4 (with the exchange epilogue's 81 registers two workgroups of ten wavefronts still share a CU).
// Measured with 2 (same box, interleaved): config 3 0.3427 against 0.3422 ms, config 2 0.3117 against 0.3108 -- two workgroups per CU already
// overlap one's address-path time with the other's arithmetic; the loaders pay where ONE workgroup owns the CU (the tiles below).  Off.
template <int AMODE, int BMODE, bool X = false>
__global__ __launch_bounds__(512 + 64 * kRingLoaders) void gemm_ring_kernel(GemmArgs g) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_map) xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, bx, by);
  gemm_ring_tile<AMODE, BMODE, X, kRingLoaders>(g, bx, by, gridDim.y);
}
// The exchange instance for launches of at most ONE workgroup per CU (layer 2 on a 128-CU share, layer 1 on the whole chip): four loader
// wavefronts.  Same box, interleaved, loaders on every exchange launch: config 2 0.3067 against 0.3098 ms (its 256-tile launches are one per
// CU); config 3 lost (two workgroups of twelve wavefronts no longer share a CU at 81 registers) -- hence per launch.
constexpr int kRingLoadersOnePerCu = 4;
template <int BMODE>
__global__ __launch_bounds__(512 + 64 * kRingLoadersOnePerCu) void gemm_ring_x1_kernel(GemmArgs g) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_map) xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, bx, by);
  gemm_ring_tile<0, BMODE, true, kRingLoadersOnePerCu>(g, bx, by, gridDim.y);
}

template <int AMODE, int BMODE, int BF = 0, int BK = kLdsBKMax, bool X = false>
__global__ __launch_bounds__(512) void gemm_lds_kernel(GemmArgs g) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_map) xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, bx, by);
  gemm_lds_tile<AMODE, BMODE, BF, BK, X>(g, bx, by, gridDim.y);
}

// The grouped launch with 64x64 LDS tiles: the weight-gradient GEMMs read both operands k-major (lin[k][i], dz[k][j]), which
// is exactly the LDS image, so a tile is staged by plain float4 copies and every operand element leaves L2 once per 64x64
// tile -- half the L2 -> CU traffic of the 32x32 fragment kernel (410 MB per step at batch 1024), which is what bounded it.
template <int AMODE, int BMODE, int BF = 0>
__global__ __launch_bounds__(512) void gemm_lds_grouped_kernel(GemmGroup grp) {
  int p = 0;
  while (p + 1 < grp.n && (int)blockIdx.x >= grp.tile_start[p + 1]) ++p;
  const GemmArgs& g = grp.g[p];
  const int local = blockIdx.x - grp.tile_start[p];
  const int ntx = (g.N + 63) / 64, nty = (g.M + 63) / 64;
  int bx = local % ntx, by = local / ntx;
  if (g.xcd_map && (grp.tile_start[p] & 7) == 0) xcd_tile(local, ntx, nty, bx, by);
  gemm_lds_tile<AMODE, BMODE, BF, kLdsBKMax>(g, bx, by, nty);
}


// ---------------------------------------------------------------------------------------------
// The weight-gradient tile of arithmetic mode 2: 128 x 128 outputs per workgroup, TWO accumulators per wavefront.
// The 64 x 64 ring tile above issues ~75 instructions per wavefront and k-step for 3 matrix instructions (one A and one B fragment
// split per 32 x 32 x 16 product) and was bound by that (its probes: 56 us with the arithmetic compiled out, 64 us with the operand
// traffic compiled out, 76 us whole, on 128 CUs).  Here wavefront (ri, cj) owns a 32 x 64 strip: per 16-deep chunk ONE A fragment is
// split and multiplies TWO B fragments -- 36 split instructions and 24 LDS reads for 6 matrix instructions --, every wavefront takes the
// whole 32-deep k-step (no K halves to add up afterwards), a k-step moves 32 KB for four times the 64 x 64 tile's arithmetic (half the
// operand traffic per flop), and the accumulators go straight to memory (the epilogue of a weight gradient is a store).  Both operands
// k-major (lin[k][i], dz[k][j]); ring of three k-steps = 96 KB, one workgroup per CU; 105 tiles for the reference's model.
// ---------------------------------------------------------------------------------------------
constexpr int kR128Stages = 3;
constexpr int kR128Tile = 32 * 128;  // floats of one operand tile of one k-step (16 KB)
constexpr size_t ring128_bytes() { return (size_t)(kR128Stages * 2 * kR128Tile) * sizeof(float); }

__device__ __forceinline__ void gemm_ring128_tile(const GemmArgs& g, const int bx, const int by) {
  constexpr int TS = 128, S = kR128Stages;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][A | B][32 k][128]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int ri = wave >> 1, cj = wave & 1;
  const int m0 = by * TS, n0 = bx * TS;
  const int m_real = g.ones_row ? g.M - 1 : g.M;
  const int nst = (g.K + 31) >> 5;
  const float rsa = scale_from_amax(g.sa_dyn, g.sa), rsb = scale_from_amax(g.sb_dyn, g.sb);
  // DMA: an instruction moves 2 k rows of 128 floats; wavefront w moves k rows 4 w .. 4 w + 3 of both operands (two instructions each)
  const int kl = 4 * wave + (lane >> 5), il = (lane & 31) * 4;
  const float* pa = g.A + (size_t)kl * g.lda + min(m0 + il, m_real - 4);
  const float* pb = g.B + (size_t)kl * g.ldb + min(n0 + il, g.N - 4);
  const size_t a2 = (size_t)2 * g.lda, b2 = (size_t)2 * g.ldb, a32 = (size_t)32 * g.lda, b32 = (size_t)32 * g.ldb;
  const bool k_tail = (g.K & 31) != 0;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds) + (unsigned)wave * 2048u;  // this wavefront's 4 k rows of an A tile
  auto issue = [&](const int t) {
    const unsigned slot = lds0 + (unsigned)(t % S) * (2u * kR128Tile * 4u);
    const bool z0 = k_tail && 32 * t + kl >= g.K, z1 = k_tail && 32 * t + kl + 2 >= g.K;
    dma16(z0 ? g.zeros : pa, slot);
    dma16(z1 ? g.zeros : pa + a2, slot + 1024u);
    dma16(z0 ? g.zeros : pb, slot + kR128Tile * 4u);
    dma16(z1 ? g.zeros : pb + b2, slot + kR128Tile * 4u + 1024u);
    pa += a32;
    pb += b32;
  };
  const int pre = min(S - 1, nst);
  for (int t = 0; t < pre; ++t) issue(t);
  const int il_ones = (g.ones_row && g.M - 1 >= m0 && g.M - 1 < m0 + TS) ? g.M - 1 - m0 : -1;
  f32x16 acc0, acc1;
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc0[q] = 0.0f; acc1[q] = 0.0f; }
  auto k_loop = [&](auto unit_a) {
  constexpr bool UA = decltype(unit_a)::value;
  for (int t = 0; t < nst; ++t) {
    const int ahead = min(t + S - 2, nst - 1) - t;  // k-steps requested beyond t: four DMA instructions each, completed in order
    if (ahead >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* At = lds + (t % S) * 2 * kR128Tile;
    if (il_ones >= 0 && lane < 4) At[(4 * wave + lane) * TS + il_ones] = 1.0f;
    __syncthreads();
#if !defined(LIPASR_RING_PROBE) || LIPASR_RING_PROBE != 2
    if (t + S - 1 < nst) issue(t + S - 1);
#endif
    const float* Bt = At + kR128Tile;
#if !defined(LIPASR_RING_PROBE) || LIPASR_RING_PROBE != 1
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* qa = At + (16 * c + 8 * hh) * TS + 32 * ri + r;
      const float* qb = Bt + (16 * c + 8 * hh) * TS + 64 * cj + r;
      float av[8], b0[8], b1[8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        av[s8] = qa[s8 * TS];
        b0[s8] = qb[s8 * TS];
        b1[s8] = qb[s8 * TS + 32];
      }
      f16x8 ah, al, bh, bl;
      split8<UA>(av, rsa, ah, al);
      split8<false>(b0, rsb, bh, bl);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc0, 0, 0, 0);
      split8<false>(b1, rsb, bh, bl);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
    }
#else
    (void)Bt;
#endif
  }
  };
  if (rsa == 1.0f) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  const float un = 1.0f / (rsa * rsb);
  const int gn0 = n0 + 64 * cj + r, gn1 = gn0 + 32;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int gm = m0 + 32 * ri + (q & 3) + 8 * (q >> 2) + 4 * hh;
    if (gm >= g.M) continue;
    float* crow = (g.ones_row && gm == g.M - 1) ? g.extra_out : g.C + (size_t)gm * g.ldc;
    if (gn0 < g.N) crow[gn0] = acc0[q] * un;
    if (gn1 < g.N) crow[gn1] = acc1[q] * un;
  }
}

// ---------------------------------------------------------------------------------------------
// The same 128 x 128 tile with a SPLIT PASS (round 5, after the counters): in the tile above every wavefront splits the fragments it
// multiplies -- the A fragment of a strip is split by both wavefronts that share it, a B fragment by all four -- and the launch was
// bound by that instruction stream (SQ counters on 128 CUs: 138 vector instructions per wavefront and k-step at ~6 cycles each, the
// wavefronts 39 % issuing / 25 % stalled on issue / 36 % parked at the barrier, the matrix pipe 16 % busy).  Here a k-step's fp32
// tile is split ONCE: each thread takes one 8-deep group of A and one of B from the ring slot (unit-stride ds_read_b32 down the
// k rows), splits them and writes the fp16 hi / lo planes K-CONTIGUOUS into a second, double-buffered region ([row][32 k] halves, the
// four 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3); the matrix pass reads a fragment as ONE ds_read_b128 per plane and
// issues no vector arithmetic at all.  Per wavefront and k-step: 16 + 12 LDS reads, ~30 vector instructions, 12 matrix instructions.
// One barrier per k-step still: iteration t splits k-step t (landed: its DMA was issued two iterations ago) while it multiplies
// k-step t - 1 (split in the iteration before), and re-issues the ring slot the previous split pass emptied.
// LDS: 3 x 32 KB ring + 2 x 32 KB planes = 160 KB, the whole CU (gfx950's addressable maximum).
// ---------------------------------------------------------------------------------------------
#ifndef LIPASR_R128_LOADERS
#define LIPASR_R128_LOADERS 4
#endif
constexpr int kR128Loaders = LIPASR_R128_LOADERS;  // 1, 2 or 4
constexpr int kR128PlaneBytes = 128 * 64;  // one fp16 plane of one operand: 128 rows x 32 k
constexpr size_t ring128s_bytes() { return ring128_bytes() + (size_t)2 * 4 * kR128PlaneBytes; }

__device__ __forceinline__ void gemm_ring128s_tile(const GemmArgs& g, const int bx, const int by) {
  constexpr int TS = 128, S = kR128Stages;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][A | B][32 k][128] fp32, then [2][A hi | A lo | B hi | B lo][128][32] fp16
  char* const planes = reinterpret_cast<char*>(lds) + ring128_bytes();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int ri = wave >> 1, cj = wave & 1;
  const int m0 = by * TS, n0 = bx * TS;
  const int m_real = g.ones_row ? g.M - 1 : g.M;
  const int nst = (g.K + 31) >> 5;
  const float rsa = scale_from_amax(g.sa_dyn, g.sa), rsb = scale_from_amax(g.sb_dyn, g.sb);
  const int il_ones = (g.ones_row && g.M - 1 >= m0 && g.M - 1 < m0 + TS) ? g.M - 1 - m0 : -1;  // the all-ones row of op(A) (bias gradients): patched into the slot
  // kR128Loaders extra wavefronts are LOADERS: they issue the 32 DMA instructions of a k-step (16 pieces of two k rows per operand) while the
  // eight others split and multiply.  With every wavefront issuing its own four pieces right behind the barrier each of them sat ~800
  // cycles of a ~2800-cycle k-step in the address path (s_memtime), the vector and matrix pipes idle meanwhile; a single wavefront gets a
  // piece accepted every ~130 cycles and the path itself takes ~64 per piece (16 B per cycle and CU), so it takes two to keep it busy.
  if (wave >= 8) {
    const int L = wave - 8;
    constexpr int NQ = 32 / kR128Loaders;  // pieces per loader and k-step: piece q -> operand q / (NQ / 2), piece index j = kR128Loaders (q % (NQ / 2)) + L
    const float* src[NQ];
    int krow[NQ];
    unsigned dst[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int op = q / (NQ / 2), j = kR128Loaders * (q % (NQ / 2)) + L;
      krow[q] = 2 * j + (lane >> 5);
      const int c = (lane & 31) * 4;
      src[q] = op ? g.B + (size_t)krow[q] * g.ldb + min(n0 + c, g.N - 4) : g.A + (size_t)krow[q] * g.lda + min(m0 + c, m_real - 4);
      dst[q] = (unsigned)op * (unsigned)(kR128Tile * 4) + (unsigned)j * 1024u;
    }
    const size_t a32 = (size_t)32 * g.lda, b32 = (size_t)32 * g.ldb;
    const bool k_tail = (g.K & 31) != 0;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    auto issue = [&](const int t) {
      const unsigned slot = lds0 + (unsigned)(t % S) * (2u * kR128Tile * 4u);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
#if defined(LIPASR_R128_PROBE) && LIPASR_R128_PROBE == 3   // (timing probes, never shipped: 1 no arithmetic, 2 no operand traffic after the prologue, 3 every DMA from one hot line)
        dma16(g.zeros, slot + dst[q]);
#else
        dma16((k_tail && 32 * t + krow[q] >= g.K) ? g.zeros : src[q], slot + dst[q]);
#endif
        src[q] += q < NQ / 2 ? a32 : b32;
      }
    };
    for (int t = 0; t < min(2, nst); ++t) issue(t);
    for (int t = 0; t <= nst; ++t) {
      if (t < nst) {  // this loader's pieces of k-step t have landed (those of t + 1 may still be in flight)
        if (t + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (il_ones >= 0 && lane < NQ) {  // the ones row, in the k rows this loader brought: A pieces j = kR128Loaders (lane / 2) + L, row lane & 1
          const int k = 2 * (kR128Loaders * (lane >> 1) + L) + (lane & 1);
          lds[(t % S) * 2 * kR128Tile + k * TS + il_ones] = 1.0f;
        }
      }
      __syncthreads();
#if !defined(LIPASR_R128_PROBE) || LIPASR_R128_PROBE != 2
      if (t + 2 < nst) issue(t + 2);  // into the slot the split pass of t - 1 emptied
#endif
    }
    return;  // (the barriers count the wavefronts that are left; there are none behind the loop)
  }
  // split pass: this thread's group = rows k = 8 sc .. 8 sc + 7 of column si, of A and of B
  const int si = 64 * (wave & 1) + lane, sc = wave >> 1;
  const unsigned sp_off = (unsigned)si * 64u + (unsigned)((sc ^ ((si >> 2) & 3)) << 4);
  // matrix pass: fragment (row, chunk c = 2 cc + hh) of a plane
  const int row_a = 32 * ri + r, row_b = 64 * cj + r;
  unsigned off_a[2], off_b[2];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    off_a[cc] = (unsigned)row_a * 64u + (unsigned)(((2 * cc + hh) ^ ((row_a >> 2) & 3)) << 4);
    off_b[cc] = (unsigned)row_b * 64u + (unsigned)(((2 * cc + hh) ^ ((row_b >> 2) & 3)) << 4);
  }
  f32x16 acc0, acc1;
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc0[q] = 0.0f; acc1[q] = 0.0f; }
  auto k_loop = [&](auto unit_a) {
    constexpr bool UA = decltype(unit_a)::value;
    for (int t = 0; t <= nst; ++t) {
      __syncthreads();  // k-step t is in its ring slot (the loaders waited for it); the split pass of t - 1 and the matrix pass of t - 2 are over everywhere
#if !defined(LIPASR_R128_PROBE) || (LIPASR_R128_PROBE != 1 && LIPASR_R128_PROBE != 3)
      if (t < nst) {
        const float* Ra = lds + (t % S) * 2 * kR128Tile + (8 * sc) * TS + si;
        char* P = planes + (t & 1) * 4 * kR128PlaneBytes + sp_off;
        float av[8], bv[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          av[s8] = Ra[s8 * TS];
          bv[s8] = Ra[kR128Tile + s8 * TS];
        }
        f16x8 h, l;
        split8<UA>(av, rsa, h, l);
        *reinterpret_cast<f16x8*>(P) = h;
        *reinterpret_cast<f16x8*>(P + kR128PlaneBytes) = l;
        split8<false>(bv, rsb, h, l);
        *reinterpret_cast<f16x8*>(P + 2 * kR128PlaneBytes) = h;
        *reinterpret_cast<f16x8*>(P + 3 * kR128PlaneBytes) = l;
      }
      if (t >= 1) {
        const char* P = planes + ((t - 1) & 1) * 4 * kR128PlaneBytes;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const f16x8 ah = *reinterpret_cast<const f16x8*>(P + off_a[cc]);
          const f16x8 al = *reinterpret_cast<const f16x8*>(P + kR128PlaneBytes + off_a[cc]);
          const f16x8 b0h = *reinterpret_cast<const f16x8*>(P + 2 * kR128PlaneBytes + off_b[cc]);
          const f16x8 b0l = *reinterpret_cast<const f16x8*>(P + 3 * kR128PlaneBytes + off_b[cc]);
          const f16x8 b1h = *reinterpret_cast<const f16x8*>(P + 2 * kR128PlaneBytes + off_b[cc] + 32 * 64);
          const f16x8 b1l = *reinterpret_cast<const f16x8*>(P + 3 * kR128PlaneBytes + off_b[cc] + 32 * 64);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc1, 0, 0, 0);
        }
      }
#endif
    }
  };
  if (rsa == 1.0f) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  const float un = 1.0f / (rsa * rsb);
  const int gn0 = n0 + 64 * cj + r, gn1 = gn0 + 32;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int gm = m0 + 32 * ri + (q & 3) + 8 * (q >> 2) + 4 * hh;
    if (gm >= g.M) continue;
    float* crow = (g.ones_row && gm == g.M - 1) ? g.extra_out : g.C + (size_t)gm * g.ldc;
    if (gn0 < g.N) crow[gn0] = acc0[q] * un;
    if (gn1 < g.N) crow[gn1] = acc1[q] * un;
  }
}

// the grouped weight-gradient launch in arithmetic mode 2: every problem that is ring_legal on the LDS-DMA ring tile, the others
// (the 64 x 10 output layer: its extent is no multiple of 4) on the register-staged tile
__global__ __launch_bounds__(512 + 64 * kR128Loaders) void gemm_ring_grouped_kernel(GemmGroup grp) {
  int p = 0;
  while (p + 1 < grp.n && (int)blockIdx.x >= grp.tile_start[p + 1]) ++p;
  const GemmArgs& g = grp.g[p];
  const int local = blockIdx.x - grp.tile_start[p];
  const int ts = g.ring >= 2 ? 128 : 64;
  const int ntx = (g.N + ts - 1) / ts, nty = (g.M + ts - 1) / ts;
  int bx = local % ntx, by = local / ntx;
  if (g.xcd_map && (grp.tile_start[p] & 7) == 0) xcd_tile(local, ntx, nty, bx, by);
  if (g.ring == 2) { gemm_ring128s_tile(g, bx, by); return; }
  if (threadIdx.x >= 512) return;  // (the loader wavefronts of the split-pass tile: the other tiles are eight wavefronts)
  if (g.ring == 3) gemm_ring128_tile(g, bx, by);
  else if (g.ring) gemm_ring_tile<1, 1>(g, bx, by, nty);
  else gemm_lds_tile<1, 1, 2, kLdsBKMax>(g, bx, by, nty);
}

// ---------------------------------------------------------------------------------------------
// The exchange-epilogue GEMMs of arithmetic mode 2 on 128 x 64 tiles (round 5): forward (A = activations, row-major; B = kernel,
// k-major) and input-gradient (B = kernel, K-contiguous) launches whose 64 x 64 tiling would put two or more workgroups on every CU
// of the plan's share.  What bounds the ring kernels on a CU share is the LDS-DMA fill rate of a CU (~32 GB/s: probe 3 of the
// weight-gradient tile above), so what counts is bytes per CU: a 128 x 64 tile moves 24 KB per k-step for the work of two 64 x 64
// tiles (32 KB).  Same split pass as the weight-gradient tile: every thread splits one 8-deep group of A (and the first 256 threads
// one of B) from the ring slot into K-contiguous fp16 planes, wavefront (ri, cj) multiplies the 32 x 32 output block (32 ri, 32 cj)
// over the whole k-step from four ds_read_b128 per 16-deep chunk; no K halves to add up.  Ring of three k-steps (72 KB) + two plane
// buffers (48 KB) + the epilogue's statistics: one workgroup per CU.  The epilogue is lds_tile_epilogue's exchange branch on four
// 32-row passes: the workgroup contributes ONE row tile of 128 rows to its column block's exchange.
// ---------------------------------------------------------------------------------------------
constexpr int kR2TileA = 128 * 32, kR2TileB = 64 * 32;            // floats of one k-step's operand tiles
constexpr int kR2Slot = kR2TileA + kR2TileB;                       // 24 KB
constexpr int kR2Stages = 3;
constexpr int kR2PlaneA = 128 * 64, kR2PlaneB = 64 * 64;           // bytes of one fp16 plane
constexpr int kR2Planes = 2 * kR2PlaneA + 2 * kR2PlaneB;           // one buffer: A hi | A lo | B hi | B lo (24 KB)
constexpr size_t ring2_bytes() { return (size_t)kR2Stages * kR2Slot * sizeof(float) + 2 * (size_t)kR2Planes + (size_t)8 * 16 * 8 * sizeof(float); }
#ifndef LIPASR_R2_LOADERS
#define LIPASR_R2_LOADERS 4
#endif
constexpr int kR2Loaders = LIPASR_R2_LOADERS;  // 1, 2, 3, 4, 6 or 8: divides the 24 pieces of a k-step

template <int BMODE>
__device__ __forceinline__ void gemm_ring2_tile(const GemmArgs& g, const int bx, const int by, const int n_row_tiles) {
  constexpr int S = kR2Stages;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [S][A 128x32 | B 64x32] fp32, [2] plane buffers, stat
  char* const planes = reinterpret_cast<char*>(lds + S * kR2Slot);
  float* const stat = reinterpret_cast<float*>(planes + 2 * kR2Planes);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int ri = wave >> 1, cj = wave & 1;
  const int m0 = by * 128, n0 = bx * 64;
  const int nst = (g.K + 31) >> 5;
  const unsigned xtag = xc_tag<64>(XcView{g.xc_gran, g.xc_ctrl, g.xc_err, g.xc_rt_max}, bx);
  const float rsa = scale_from_amax(g.sa_dyn, g.sa), rsb = scale_from_amax(g.sb_dyn, g.sb);
  if (g.amax_zero && bx == 0 && by == 0) amax_clear(g.amax_zero);
  // kR2Loaders extra wavefronts are loaders (see the weight-gradient tile): the 24 pieces of a k-step -- 16 of 8 rows of the A tile (the eight
  // 16-byte chunks of a row XOR-swizzled through the source address as in the 64 x 64 ring tile), 8 of the B tile -- dealt round-robin
  if (wave >= 8) {
    const int L = wave - 8;
    constexpr int NQ = 24 / kR2Loaders;
    const float* src[NQ];
    int koff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int j = kR2Loaders * q + L;  // piece 0 .. 23
      if (j < 16) {
        const int row = 8 * j + (lane >> 3);
        koff[q] = 4 * ((lane & 7) ^ ((row >> 1) & 7));
        src[q] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + koff[q];
      } else if (BMODE == 1) {
        koff[q] = 4 * (j - 16) + (lane >> 4);
        src[q] = g.B + (size_t)koff[q] * g.ldb + min(n0 + (lane & 15) * 4, g.N - 4);
      } else {
        const int br = 8 * (j - 16) + (lane >> 3);
        koff[q] = 4 * ((lane & 7) ^ ((br >> 1) & 7));
        src[q] = g.B + (size_t)min(n0 + br, g.N - 1) * g.ldb + koff[q];
      }
    }
    const size_t sb_step = BMODE == 1 ? (size_t)32 * g.ldb : (size_t)32;
    const bool k_tail = (g.K & 31) != 0;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    auto issue = [&](const int t) {
      const unsigned slot = lds0 + (unsigned)(t % S) * (unsigned)(kR2Slot * 4);
      const bool last = k_tail && t == nst - 1;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        dma16((last && 32 * t + koff[q] >= g.K) ? g.zeros : src[q], slot + (unsigned)(kR2Loaders * q + L) * 1024u);
        src[q] += (kR2Loaders * q + L) < 16 ? (size_t)32 : sb_step;
      }
    };
    for (int t = 0; t < min(2, nst); ++t) issue(t);
    for (int t = 0; t <= nst; ++t) {
      if (t < nst) {
        if (t + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NQ) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (t + 2 < nst) issue(t + 2);  // into the slot the split pass of t - 1 emptied
    }
    return;  // (the barriers of the epilogue count the wavefronts that are left)
  }
  // split pass: row si = 64 (w & 1) + lane, chunk sc = w >> 1 of A; wavefronts 0 .. 3 also row / column `lane`, chunk w of B
  const int si = 64 * (wave & 1) + lane, sc = wave >> 1;
  const unsigned spa = (unsigned)si * 64u + (unsigned)((sc ^ ((si >> 2) & 3)) << 4);
  const unsigned spb = (unsigned)lane * 64u + (unsigned)(((wave & 3) ^ ((lane >> 2) & 3)) << 4);
  // matrix pass: fragment (row, chunk 2 cc + hh)
  const int row_a = 32 * ri + r, row_b = 32 * cj + r;
  unsigned off_a[2], off_b[2];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    off_a[cc] = (unsigned)row_a * 64u + (unsigned)(((2 * cc + hh) ^ ((row_a >> 2) & 3)) << 4);
    off_b[cc] = (unsigned)(2 * kR2PlaneA) + (unsigned)row_b * 64u + (unsigned)(((2 * cc + hh) ^ ((row_b >> 2) & 3)) << 4);
  }
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  auto k_loop = [&](auto unit_a) {
    constexpr bool UA = decltype(unit_a)::value;
    for (int t = 0; t <= nst; ++t) {
      __syncthreads();  // k-step t is in its ring slot (the loaders waited for it); the split pass of t - 1 and the matrix pass of t - 2 are over everywhere
      // Order inside a k-step (one dependent chain per wavefront, so the LDS round trips are put behind each other's shadow): the
      // split pass's raw reads and the matrix pass's first fragments are requested together, the splits run while the fragments
      // arrive, the second chunk's fragments are requested in front of the first chunk's matrix instructions.
      const float* Rs = lds + (t % S) * kR2Slot;
      const char* Pm = planes + ((t - 1) & 1) * kR2Planes;
      float va[8], vb[8];
      f16x8 ah, al, bh, bl;
      if (t < nst) {
        ring_frag<0>(Rs, si, sc >> 1, sc & 1, va);
        if (wave < 4) ring_frag<BMODE>(Rs + kR2TileA, lane, wave >> 1, wave & 1, vb);
      }
      if (t >= 1) {
        ah = *reinterpret_cast<const f16x8*>(Pm + off_a[0]);
        al = *reinterpret_cast<const f16x8*>(Pm + kR2PlaneA + off_a[0]);
        bh = *reinterpret_cast<const f16x8*>(Pm + off_b[0]);
        bl = *reinterpret_cast<const f16x8*>(Pm + kR2PlaneB + off_b[0]);
      }
      if (t < nst) {
        char* P = planes + (t & 1) * kR2Planes;
        f16x8 h, l;
        split8<UA>(va, rsa, h, l);
        *reinterpret_cast<f16x8*>(P + spa) = h;
        *reinterpret_cast<f16x8*>(P + kR2PlaneA + spa) = l;
        if (wave < 4) {
          split8<false>(vb, rsb, h, l);
          *reinterpret_cast<f16x8*>(P + 2 * kR2PlaneA + spb) = h;
          *reinterpret_cast<f16x8*>(P + 2 * kR2PlaneA + kR2PlaneB + spb) = l;
        }
      }
      if (t >= 1) {
        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(Pm + off_a[1]);
        const f16x8 al1 = *reinterpret_cast<const f16x8*>(Pm + kR2PlaneA + off_a[1]);
        const f16x8 bh1 = *reinterpret_cast<const f16x8*>(Pm + off_b[1]);
        const f16x8 bl1 = *reinterpret_cast<const f16x8*>(Pm + kR2PlaneB + off_b[1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, acc, 0, 0, 0);
      }
    }
  };
  if (rsa == 1.0f) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  // ---- epilogue: the exchange branch of lds_tile_epilogue on four 32-row passes
  constexpr int TS = 64;
  const int tcol = tid & 15, trow = tid >> 4;
  const int c4 = tcol * 4, gn = n0 + c4;
  const int gm4[4] = {m0 + trow, m0 + trow + 32, m0 + trow + 64, m0 + trow + 96};
  BnxPre xpre[4];
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) bnx_prefetch(g, gm4[p4], gn, xpre[p4]);
  const float un = 1.0f / (rsa * rsb);
  float* red = lds;  // [128][64] over the ring slots: their last reader was the split pass of k-step nst - 1, in front of the loop's last barrier
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = 32 * ri + (q & 3) + 8 * (q >> 2) + 4 * hh;
    red[row * TS + 32 * cj + r] = acc[q] * un;
  }
  __syncthreads();
  const int step = g.drop.step_dev ? *g.drop.step_dev : 0;
  float val[4][4], av[4][4], c1[4] = {0.f, 0.f, 0.f, 0.f}, c2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p4 = 0; p4 < 4; ++p4) {
    const float4 s = *reinterpret_cast<const float4*>(red + (trow + 32 * p4) * TS + c4);
    const float accv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t1, t2;
      bnx_elem(g, step, gm4[p4] < g.M && gn + e < g.N, gm4[p4], gn + e, accv[e], xpre[p4], e, val[p4][e], av[p4][e], t1, t2);
      c1[e] += t1;
      c2[e] += t2;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    c1[e] += __shfl_xor(c1[e], 16, 64); c1[e] += __shfl_xor(c1[e], 32, 64);
    c2[e] += __shfl_xor(c2[e], 16, 64); c2[e] += __shfl_xor(c2[e], 32, 64);
  }
  if (lane < 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      stat[(wave * 16 + lane) * 8 + e] = c1[e];
      stat[(wave * 16 + lane) * 8 + 4 + e] = c2[e];
    }
  }
  __syncthreads();  // (every read of `red` is done: it is carved up below)
  float* mine = red;                                       // [2][64]
  float* colp = red + 128;                                 // [2][64]
  double* sbuf = reinterpret_cast<double*>(red + 256);     // [4][128]
  double* tot = sbuf + 4 * 128;                            // [128]
  if (tid < 2 * TS) {
    const int which = tid / TS, col = tid % TS, l4 = col >> 2, e = col & 3;
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += stat[(w * 16 + l4) * 8 + which * 4 + e];
    mine[tid] = t;
  }
  __syncthreads();
  BnxLate late;
  bnx_late_load(g, gn, late);
  float mm0, mv0;
  bnx_moving_load(g, by, n0 + tid, tid < TS, mm0, mv0);
  XcView xc{g.xc_gran, g.xc_ctrl, g.xc_err, g.xc_rt_max};
  xc_exchange<512, 64>(xc, bx, by, g.Bstat < 0 ? 0 : n_row_tiles, xtag, mine, sbuf, tot, [&]() {
    if (g.epi != EPI_BIAS_RELU_BNX) return;
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
      if (gm4[p4] >= g.M) continue;
      float* crow = g.C + (size_t)gm4[p4] * g.ldc;
      if (gn + 3 < g.N && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(crow) & 15) == 0)) {
        *reinterpret_cast<float4*>(crow + gn) = make_float4(val[p4][0], val[p4][1], val[p4][2], val[p4][3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gn + e < g.N) crow[gn + e] = val[p4][e];
      }
    }
  });
  if (tid < TS) bnx_column(g, by, n0 + tid, tid, TS, tot, colp, mm0, mv0);
  __syncthreads();
  bnx_finish<4>(g, step, gm4, gn, val, av, colp, TS, c4, late);
}

template <int BMODE>
__global__ __launch_bounds__(512 + 64 * kR2Loaders) void gemm_ring2_kernel(GemmArgs g) {
  gemm_ring2_tile<BMODE>(g, blockIdx.x, blockIdx.y, gridDim.y);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// amode / bmode: 1 = the operand is k-major in memory (P[k ld + i])
static int g_ring_tile = 2;  // weight-gradient group: 2 = the 128 x 128 tile with the split pass, 3 = the 128 x 128 tile that splits per fragment (lipasr_debug_gemm_mode
                             // bit 7), 1 = the 64 x 64 ring tile (bit 6)
static int g_no_ring = 0;  // lipasr_debug_gemm_mode bit 5: arithmetic mode 2 on the register-staged tiles only (A/B knob)
static bool ring_legal(int amode, int bmode, const GemmArgs& g) {
  if (g_no_ring || g.bf16 != 2 || g.M < 64 || g.N < 64 || g.K < 32) return false;
  if ((g.K & 31) && (!g.zeros || (g.K & 3))) return false;  // a K tail needs the zero source, and whole 16-byte chunks
  if ((g.lda & 3) || (g.ldb & 3) || !aligned16(g.A) || !aligned16(g.B)) return false;
  const int m_real = g.ones_row ? g.M - 1 : g.M;
  if (amode == 1 && ((m_real & 3) || m_real < 4)) return false;
  if (bmode == 1 && ((g.N & 3) || g.N < 4)) return false;
  return true;
}

// The 128 x 64 exchange tile (gemm_ring2_tile) pays where the 64 x 64 tiling would put two workgroups on (nearly) every CU of the
// plan's share: then it halves the workgroups and moves 3/4 of the bytes per CU; with fewer tiles than that it would leave CUs idle.
// One workgroup per CU (124 KB of LDS) and every workgroup resident (the exchange): at most `cus` tiles.
static int g_ring2 = 1;  // lipasr_debug_gemm_mode bit 8 clears it (A/B knob)
static int g_ring_x1 = 1;  // bit 9 clears it: no loader-wavefront instance for exchange launches of one workgroup per CU
static long g_launch_count[2] = {0, 0};  // lipasr_debug_launch_count: 0 = launches on 128 x 64 exchange tiles, 1 = weight-gradient launches with 128 x 128 split-pass tiles
static bool use_ring2(int bmode, const GemmArgs& g) {
  if (!g_ring2 || g.bf16 != 2 || g.M < 128 || g.N < 64 || !ring_legal(0, bmode, g)) return false;
  if (bmode == 0 && g.N < 64) return false;
  int cus = g.cus;
  if (cus <= 0) { hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev); cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256; }
  const long tiles = (long)((g.M + 127) / 128) * ((g.N + 63) / 64);
  return 4 * tiles >= 3 * (long)cus && tiles <= (long)cus && (g.M + 127) / 128 <= g.xc_rt_max;
}

// arithmetic mode (GemmArgs::bf16: 0 exact fp32, 1 bf16 operands, 2 fp16 two-plane split) -> template instance
#define LP_DISPATCH_AR(ar, M) \
  do {                        \
    if ((ar) == 2) { M(2) }   \
    else if ((ar) == 1) { M(1) } \
    else { M(0) }             \
  } while (0)

// launches up to kMaxGroup weight-gradient style GEMMs (AMODE 1, BMODE 1) as one grid
static int g_group_lds = 1;  // lipasr_debug_gemm_mode bit 3 clears it: the grouped launch on 32x32 fragment tiles (round 2)

static int launch_gemm_group_tn(const GemmArgs* gs, int n, hipStream_t st) {
  int done = 0;
  while (done < n) {
    GemmGroup grp;
    memset(&grp, 0, sizeof(grp));
    // 64x64 LDS tiles when the group is large enough to fill the chip with them (measured: bench config 3, batch 1024)
    long big = 0;
    for (int k = 0; k < kMaxGroup && done + k < n; ++k)
      big += (long)((gs[done + k].N + 63) / 64) * ((gs[done + k].M + 63) / 64);
    const bool lds_tiles = g_group_lds && big >= 128 && gs[done].K >= 64;
    const int ts = lds_tiles ? 64 : 32;
    const int ar = gs[done].bf16;
    // 128 x 128 tiles (one workgroup per CU, half the operand bytes per flop, two accumulators per wavefront).  A CU takes in ~32 GB/s
    // whatever asks (LDS-DMA or register loads), so a launch is as long as its busiest CU's bytes: one 128 x 128 tile = 1 MB, two 64 x 64
    // tiles per CU = 1 MB as well but four per CU 2 MB.  The reference's model makes 105 + 4 tiles: with the split pass and the loader
    // wavefronts 38-40 us on a 128-CU share (64 x 64 ring tiles: 77) and config 2's step 0.310 against 0.316 ms on all 256 CUs, where
    // they leave 150 CUs idle -- so: whenever the tiles cover 40 % of the CUs the launch may use.
    int ring_tile = g_ring_tile;
    if (ring_tile >= 2 && lds_tiles && ar == 2) {
      long n128 = 0;
      for (int q = 0; q < kMaxGroup && done + q < n; ++q)
        if (ring_legal(1, 1, gs[done + q])) n128 += (long)((gs[done + q].N + 127) / 128) * ((gs[done + q].M + 127) / 128);
      int cus = gs[done].cus;
      if (cus <= 0) { hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev); cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256; }
      if (10 * n128 < 4 * (long)cus) ring_tile = 1;
    }
    int k = 0, tiles = 0;
    bool any_ring = false;
    for (; k < kMaxGroup && done + k < n; ++k) {
      const GemmArgs& g = gs[done + k];
      if (g.M <= 0 || g.N <= 0 || g.K <= 0) { set_error("grouped gemm: empty problem"); return LIPASR_EINVAL; }
      grp.g[k] = g;
      // arithmetic mode 2: the 128 x 128 two-accumulator ring tile for every problem it can take (g_ring_tile 1: the 64 x 64 ring tile)
      if (lds_tiles && ar == 2 && ring_legal(1, 1, g)) { grp.g[k].ring = ring_tile; any_ring = true; }
      const int tp = grp.g[k].ring >= 2 ? 128 : ts;
      grp.tile_start[k] = tiles;
      tiles += ((g.N + tp - 1) / tp) * ((g.M + tp - 1) / tp);
    }
    grp.n = k;
    grp.tile_start[k] = tiles;
    if (any_ring) {
      // dynamic LDS the kernel may ask for, set once per device: 160 KB (the split-pass tile: the whole CU) where the device grants it
      static int attr_dev_max[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); int& attr_max = attr_dev_max[attr_dev & 15];
      if (!attr_max) {
        const void* fn = reinterpret_cast<const void*>(gemm_ring_grouped_kernel);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring128s_bytes()) == hipSuccess) attr_max = (int)ring128s_bytes();
        else {
          (void)hipGetLastError();
          (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(ring_gemm_bytes(), ring128_bytes()));
          attr_max = (int)std::max(ring_gemm_bytes(), ring128_bytes());
        }
      }
      if (ring_tile == 2 && (size_t)attr_max < ring128s_bytes()) {  // (never seen on gfx950)
        ring_tile = 3;
        for (int q = 0; q < k; ++q) if (grp.g[q].ring == 2) grp.g[q].ring = 3;
      }
      const size_t lds_r = std::max(ring_gemm_bytes(), ring_tile == 2 ? ring128s_bytes() : ring_tile == 3 ? ring128_bytes() : (size_t)0);
      if (ring_tile == 2) ++g_launch_count[1];
      hipLaunchKernelGGL(gemm_ring_grouped_kernel, dim3(tiles), dim3(ring_tile == 2 ? 512 + 64 * kR128Loaders : 512), lds_r, st, grp);  // (loader wavefronts: the split-pass tile only)
    } else if (lds_tiles) {
      constexpr size_t lds_b = lds_gemm_bytes(kLdsBKMax);
      static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];  /* per device */
      if (!attr_set) {
#define LP_M(AR) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_lds_grouped_kernel<1, 1, AR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        LP_M(0) LP_M(1) LP_M(2)
#undef LP_M
        attr_set = true;
      }
#define LP_M(AR) hipLaunchKernelGGL((gemm_lds_grouped_kernel<1, 1, AR>), dim3(tiles), dim3(512), lds_b, st, grp);
      LP_DISPATCH_AR(ar, LP_M);
#undef LP_M
    } else {
      const size_t lds = (size_t)(4 * 32 * 32 + 4 * 8 * 8) * sizeof(float);
#define LP_M(AR) hipLaunchKernelGGL((gemm_f32_grouped_kernel<1, 1, 4, AR>), dim3(tiles), dim3(256), lds, st, grp);
      LP_DISPATCH_AR(ar, LP_M);
#undef LP_M
    }
    LP_LAUNCH_CHECK();
    done += k;
  }
  return LIPASR_OK;
}

static int g_xcd_map = 0;    // lipasr_debug_gemm_mode bit 4 SETS it (round 5: measured, not kept): the XCD-aware blockIdx -> tile map of xcd_tile().
                             // Same box, interleaved: config 2 0.3177 with it against 0.3148 without, config 3 0.3653 against 0.3625; the counters
                             // (TCC hit 67 % on the weight-gradient launch either way) say the operand panels are not what misses.
static int g_gemm_mode = 0;  // 0 auto, 1 split-K kernel only, 2 LDS kernel wherever it is legal (profiling knob)
static int g_split_dw0 = 0;  // lipasr_debug_gemm_mode bit 2: the first layer's weight gradient as its own launch

constexpr int kLdsMinTiles = 224;  // measured on MI355X (scratch/time_gemm.py): the LDS kernel wins from ~one tile per CU

static bool use_lds_gemm(int M, int N, int K, int min_tiles = 0) {
  if (g_gemm_mode == 1) return false;
  const bool legal = M >= 64 && N >= 64 && K >= 32;
  if (g_gemm_mode == 2) return legal;
  const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
  return legal && tiles >= (min_tiles > 0 ? min_tiles : kLdsMinTiles);
}

template <int AMODE, int BMODE, int NW, int BF>
static void launch_gemm_tb(const GemmArgs& g, hipStream_t st) {
  const dim3 grid((g.N + 31) / 32, (g.M + 31) / 32);
  const size_t lds = (size_t)(NW * 32 * 32 + 4 * 8 * 8) * sizeof(float);
  static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];  /* per device */
  if (lds > 48 * 1024 && !attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<AMODE, BMODE, NW, BF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_f32_kernel<AMODE, BMODE, NW, BF>), grid, dim3(64 * NW), lds, st, g);
}

template <int AMODE, int BMODE, int NW>
static void launch_gemm_t(const GemmArgs& g, hipStream_t st) {
#define LP_M(AR) launch_gemm_tb<AMODE, BMODE, NW, AR>(g, st);
  LP_DISPATCH_AR(g.bf16, LP_M);
#undef LP_M
}

static int launch_gemm(int amode, int bmode, const GemmArgs& g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) {
    set_error("gemm: empty problem %dx%dx%d", g.M, g.N, g.K);
    return LIPASR_EINVAL;
  }
  // Few output tiles and a long K (the dW GEMMs of the narrow layers, K = batch): 16 wavefronts split K so that
  // the serial chain of chunk loads per wavefront stays short.  Not for the *_STATS epilogues (never needed there).
  if (g.epi == EPI_BIAS_SOFTMAX_CE) {  // one 32-wide column tile, 4 wavefronts: the epilogue reduces along lanes
    if (g.N > 32) { set_error("gemm: the fused softmax epilogue needs N <= 32 (got %d)", g.N); return LIPASR_EINVAL; }
    if (amode == 0 && bmode == 1) launch_gemm_t<0, 1, 4>(g, st);
    else { set_error("gemm: the fused softmax epilogue is a forward (NN) epilogue"); return LIPASR_EINVAL; }
    LP_LAUNCH_CHECK();
    return LIPASR_OK;
  }
  if (g.epi == EPI_BIAS_RELU_BNX || g.epi == EPI_DH_BNX) {  // the exchange epilogue: forward (NN) or input-gradient (NT) GEMMs only
    if (amode != 0) { set_error("gemm: the exchange epilogue needs a row-major A operand"); return LIPASR_EINVAL; }
    if (use_ring2(bmode, g)) {  // 128 x 64 tiles, one workgroup per CU of the plan's share
      ++g_launch_count[0];
      const dim3 grid((g.N + 63) / 64, (g.M + 127) / 128);
      static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring2_bytes());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring2_bytes());
        attr_set = true;
      }
      if (bmode == 0) hipLaunchKernelGGL((gemm_ring2_kernel<0>), grid, dim3(512 + 64 * kR2Loaders), ring2_bytes(), st, g);
      else hipLaunchKernelGGL((gemm_ring2_kernel<1>), grid, dim3(512 + 64 * kR2Loaders), ring2_bytes(), st, g);
    } else if (use_lds_gemm(g.M, g.N, g.K, g.lds_min_tiles) && ring_legal(0, bmode, g)) {
      const dim3 grid((g.N + 63) / 64, (g.M + 63) / 64);
      static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_gemm_bytes());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<0, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_gemm_bytes());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_x1_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_gemm_bytes());
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_x1_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_gemm_bytes());
        attr_set = true;
      }
      int cus = g.cus;
      if (cus <= 0) { hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev); cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256; }
      if (g_ring_x1 && (long)grid.x * grid.y <= (long)cus) {  // at most one workgroup per CU: the instance with loader wavefronts
        if (bmode == 0) hipLaunchKernelGGL((gemm_ring_x1_kernel<0>), grid, dim3(512 + 64 * kRingLoadersOnePerCu), ring_gemm_bytes(), st, g);
        else hipLaunchKernelGGL((gemm_ring_x1_kernel<1>), grid, dim3(512 + 64 * kRingLoadersOnePerCu), ring_gemm_bytes(), st, g);
      } else if (bmode == 0) hipLaunchKernelGGL((gemm_ring_kernel<0, 0, true>), grid, dim3(512 + 64 * kRingLoaders), ring_gemm_bytes(), st, g);
      else hipLaunchKernelGGL((gemm_ring_kernel<0, 1, true>), grid, dim3(512 + 64 * kRingLoaders), ring_gemm_bytes(), st, g);
    } else if (use_lds_gemm(g.M, g.N, g.K, g.lds_min_tiles)) {
      const dim3 grid((g.N + 63) / 64, (g.M + 63) / 64);
      constexpr size_t lds_b = lds_gemm_bytes(kLdsBKMax);
      static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];
      if (!attr_set) {
#define LP_M(AR)                                                                                                                                   \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_lds_kernel<0, 0, AR, kLdsBKMax, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b); \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_lds_kernel<0, 1, AR, kLdsBKMax, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        LP_M(0) LP_M(1) LP_M(2)
#undef LP_M
        attr_set = true;
      }
      if (bmode == 0) {
#define LP_M(AR) hipLaunchKernelGGL((gemm_lds_kernel<0, 0, AR, kLdsBKMax, true>), grid, dim3(512), lds_b, st, g);
        LP_DISPATCH_AR(g.bf16, LP_M);
#undef LP_M
      } else {
#define LP_M(AR) hipLaunchKernelGGL((gemm_lds_kernel<0, 1, AR, kLdsBKMax, true>), grid, dim3(512), lds_b, st, g);
        LP_DISPATCH_AR(g.bf16, LP_M);
#undef LP_M
      }
    } else {
      const dim3 grid((g.N + 31) / 32, (g.M + 31) / 32);
      const size_t lds = (size_t)(4 * 32 * 32 + 4 * 8 * 8) * sizeof(float);
      if (bmode == 0) {
#define LP_M(AR) hipLaunchKernelGGL((gemm_f32_kernel<0, 0, 4, AR, true>), grid, dim3(256), lds, st, g);
        LP_DISPATCH_AR(g.bf16, LP_M);
#undef LP_M
      } else {
#define LP_M(AR) hipLaunchKernelGGL((gemm_f32_kernel<0, 1, 4, AR, true>), grid, dim3(256), lds, st, g);
        LP_DISPATCH_AR(g.bf16, LP_M);
#undef LP_M
      }
    }
    LP_LAUNCH_CHECK();
    return LIPASR_OK;
  }
  if (use_lds_gemm(g.M, g.N, g.K, g.lds_min_tiles) && ring_legal(amode, bmode, g)) {
    const dim3 grid((g.N + 63) / 64, (g.M + 63) / 64);
#define LP_RING(A_, B_)                                                                                                                 \
  {                                                                                                                                     \
    static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];   \
    if (!attr_set) {                                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<A_, B_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_gemm_bytes()); \
      attr_set = true;                                                                                                                  \
    }                                                                                                                                   \
    hipLaunchKernelGGL((gemm_ring_kernel<A_, B_>), grid, dim3(512 + 64 * kRingLoaders), ring_gemm_bytes(), st, g);                                          \
  }
    if (amode == 0 && bmode == 0) LP_RING(0, 0) else if (amode == 0 && bmode == 1) LP_RING(0, 1)
    else if (amode == 1 && bmode == 0) LP_RING(1, 0) else LP_RING(1, 1)
#undef LP_RING
    LP_LAUNCH_CHECK();
    return LIPASR_OK;
  }
  if (use_lds_gemm(g.M, g.N, g.K, g.lds_min_tiles)) {
    const dim3 grid((g.N + 63) / 64, (g.M + 63) / 64);
    constexpr size_t lds_b = lds_gemm_bytes(kLdsBKMax);
#define LP_LDS(A_, B_)                                                                                                                  \
  {                                                                                                                                     \
    static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];   \
    if (!attr_set) {                                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_lds_kernel<A_, B_, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b); \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_lds_kernel<A_, B_, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b); \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_lds_kernel<A_, B_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b); \
      attr_set = true;                                                                                                                  \
    }                                                                                                                                   \
    if (g.bf16 == 2) hipLaunchKernelGGL((gemm_lds_kernel<A_, B_, 2>), grid, dim3(512), lds_b, st, g);                                   \
    else if (g.bf16 == 1) hipLaunchKernelGGL((gemm_lds_kernel<A_, B_, 1>), grid, dim3(512), lds_b, st, g);                              \
    else hipLaunchKernelGGL((gemm_lds_kernel<A_, B_, 0>), grid, dim3(512), lds_b, st, g);                                               \
  }
    if (amode == 0 && bmode == 0) LP_LDS(0, 0) else if (amode == 0 && bmode == 1) LP_LDS(0, 1)
    else if (amode == 1 && bmode == 0) LP_LDS(1, 0) else LP_LDS(1, 1)
#undef LP_LDS
    LP_LAUNCH_CHECK();
    return LIPASR_OK;
  }
  const long tiles = (long)((g.N + 31) / 32) * ((g.M + 31) / 32);
  const bool deep = tiles <= 192 && g.K >= 512 && g.epi != EPI_BIAS_RELU_STATS && g.epi != EPI_DH_STATS && g.epi != EPI_BIAS_RELU_BNX &&
                    g.epi != EPI_DH_BNX;
  if (deep) {
    if (amode == 0 && bmode == 0) launch_gemm_t<0, 0, 16>(g, st);
    else if (amode == 0 && bmode == 1) launch_gemm_t<0, 1, 16>(g, st);
    else if (amode == 1 && bmode == 0) launch_gemm_t<1, 0, 16>(g, st);
    else launch_gemm_t<1, 1, 16>(g, st);
  } else {
    if (amode == 0 && bmode == 0) launch_gemm_t<0, 0, 4>(g, st);
    else if (amode == 0 && bmode == 1) launch_gemm_t<0, 1, 4>(g, st);
    else if (amode == 1 && bmode == 0) launch_gemm_t<1, 0, 4>(g, st);
    else launch_gemm_t<1, 1, 4>(g, st);
  }
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

// row tiles of the *_STATS epilogues: must follow the kernel choice of launch_gemm for the same (M, N, K)
static int stats_row_tiles(int M, int N, int K, int min_tiles) { return use_lds_gemm(M, N, K, min_tiles) ? (M + 63) / 64 : (M + 31) / 32; }

// The exchange epilogue (EPI_*_BNX) spins until every row tile of its column block has published: every workgroup of the launch
// must be able to be resident at the same time.  Workgroups per CU = the occupancy query capped by the SGPR admission rule
// (blocks_per_cu), times the CUs the plan's stream may use.  The row tiles must also fit the
// granule regions.  Anything else takes the launch chain (GEMM + apply kernel).
template <typename K>
static int blocks_per_cu(K kernel, int threads, size_t lds) {
  int occ = 0;
  // the query answers 0 for a dynamic LDS size the kernel has not been allowed yet (the launchers raise the limit at a kernel's FIRST
  // launch -- which, for the exchange instances, only happens once this function has said they fit: in a fresh process the first
  // mode-2 training step then took the launch chain for every LDS-tiled layer, found by the launch-count test hook)
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
  // MI355X_MICROARCH.md (residency): the hardware admits 256-thread blocks up to floor(800 / (ceil(sgpr / 16) 16 + 16)) per CU,
  // which the query does not know about (it can be one high).  These kernels use 102-106 SGPRs (GemmArgs is a large by-value
  // argument): 800 / 128 = 6 blocks of four wavefronts = 24 wavefronts per CU.
  const int by_sgpr = 24 / (threads / 64);
  return std::min(occ, by_sgpr);
}

static bool bnx_fits(const lipasr_mlp* m, bool forward, int M, int N, int K) {
  if (!m->fuse_bn || !m->xc_gran) return false;
  const bool lds_k = use_lds_gemm(M, N, K, m->lds_min_tiles);
  const int ts = lds_k ? 64 : 32;
  const long row_tiles = (M + ts - 1) / ts, tiles = row_tiles * ((N + ts - 1) / ts);
  if (row_tiles > m->xc_rt_max || row_tiles > 64) return false;
  // [forward | backward][fragment | LDS kernel][fp32 | bf16 operands], filled on first use (one device per process in practice;
  // the kernels' resource use does not depend on the device)
  static int per_cu[2][2][3] = {{{-1, -1, -1}, {-1, -1, -1}}, {{-1, -1, -1}, {-1, -1, -1}}};
  const int ar = m->compute_bf16;
  int& pc = per_cu[forward ? 0 : 1][lds_k ? 1 : 0][ar];
  if (pc < 0) {
    const size_t lds_f = (size_t)(4 * 32 * 32 + 4 * 8 * 8) * sizeof(float);
#define LP_M(AR)                                                                                                              \
  if (forward) pc = lds_k ? blocks_per_cu(gemm_lds_kernel<0, 1, AR, kLdsBKMax, true>, 512, lds_gemm_bytes(kLdsBKMax))       \
                          : blocks_per_cu(gemm_f32_kernel<0, 1, 4, AR, true>, 256, lds_f);                                   \
  else pc = lds_k ? blocks_per_cu(gemm_lds_kernel<0, 0, AR, kLdsBKMax, true>, 512, lds_gemm_bytes(kLdsBKMax))               \
                  : blocks_per_cu(gemm_f32_kernel<0, 0, 4, AR, true>, 256, lds_f);
    LP_DISPATCH_AR(ar, LP_M);
#undef LP_M
  }
  int per = pc;
  if (ar == 2 && lds_k) {  // the launch may take the LDS-DMA ring instance (68 KB of LDS): the smaller of the two answers
    static int ring_pc[2] = {-1, -1};
    int& rp = ring_pc[forward ? 0 : 1];
    if (rp < 0) rp = forward ? blocks_per_cu(gemm_ring_kernel<0, 1, true>, 512 + 64 * kRingLoaders, ring_gemm_bytes()) : blocks_per_cu(gemm_ring_kernel<0, 0, true>, 512 + 64 * kRingLoaders, ring_gemm_bytes());
    per = std::min(per, rp);
  }
  const int cus = m->cu_budget > 0 ? std::min(m->cu_budget, m->n_cus) : m->n_cus;
  return per > 0 && tiles <= (long)per * cus;
}

static GemmArgs gemm_args(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                          int epi) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.epi = epi;
  g.xcd_map = g_xcd_map;
  g.sa = g.sb = 1.0f;
  return g;
}

// arithmetic mode of a plan's GEMM + the fp16 range scales of mode 2 by operand kind.  Powers of two (exact): activations and
// features unscaled (|x| < 65504; BatchNorm outputs are O(1): their low plane is a normal fp16 number down to |x| = 0.12 and loses
// bits gradually below, on values that weigh little in a sum), kernels x 2^12 (|w| < 16), gradients x 2^8 / 2^floor(log2 g0) where g0 is the
// size of the gradient at the network's output (the loss gradient is <= 1 / batch: a batch of 1024 gives 2^18, and room for the
// gradient to grow 256-fold on its way down).  A value outside its range becomes inf in the fp16 conversion and the loss NaN --
// loud, not silently wrong.  The low plane keeps its full 11 bits while |x| scale >= 2^-3 and degrades gradually below (fp16
// subnormals): with the first gradient scale tried, a fixed 2^14, the bias gradients of a 1024-row batch (sums of 1024 terms of
// 1e-6 that cancel to 5e-7) came out 9e-5 off; scaled by the batch they are inside the exact mode's 5e-5.
enum OperandKind { OP_ACT = 0, OP_WEIGHT = 1, OP_GRAD = 2 };
static float grad_scale_for(float g0) {  // g0: bound of the gradient at the logits (inv_batch, or 1 for a caller-given upstream vector)
  int e = 0;
  (void)frexpf(g0 > 0.0f ? g0 : 1.0f, &e);  // g0 = f 2^e, f in [0.5, 1)
  return ldexpf(1.0f, 8 - e);               // 2^8 / 2^e >= 2^8 / (2 g0)
}
static void set_arith(GemmArgs& g, const lipasr_mlp* m, int kind_a, int kind_b, float g0 = 1.0f, bool training = true) {
  // Mode 2 is a TRAINING arithmetic: there the activations are BatchNorm outputs (bounded by construction) and the gradients carry
  // their own measured scale.  Inference-mode activations have no such bound (BatchNorm with moving statistics that have not
  // adapted yet passes 1e4-sized values: the fp16 conversion overflowed in the first suite run) -- predict / attacks / class
  // gradients stay on the exact fp32 chains, so every logit-parity statement is about exact fp32 whatever the training mode.
  g.bf16 = (m->compute_bf16 == 2 && !training) ? 0 : m->compute_bf16;
  g.zeros = reinterpret_cast<const float*>(m->xc_ctrl + 4);  // words 4 .. 7 of the control block: never written
  g.cus = m->cu_budget > 0 ? std::min(m->cu_budget, m->n_cus) : m->n_cus;
  const float sc[3] = {1.0f, 4096.0f, grad_scale_for(g0)};
  g.sa = sc[kind_a];
  g.sb = sc[kind_b];
}

// ---------------------------------------------------------------------------------------------
// BatchNorm / dropout apply kernels.  The column statistics arrive as per-row-tile partial sums from
// the producing GEMM's epilogue, so these are plain 2-D elementwise kernels: grid = (column strips of
// 128, row chunks of 32), 256 threads = 32 float4 column lanes x 8 row lanes, each workgroup re-sums
// the (few) partials of its columns in fp64 in its prologue.  No cross-workgroup step, fixed order.
// ---------------------------------------------------------------------------------------------
constexpr int kApplyRows = 32;

struct ColLane {
  int j;      // first of this lane's 4 columns
  bool vec;   // float4 access is legal
  int N;
};

__device__ __forceinline__ void ld4(const float* __restrict__ p, size_t row_off, const ColLane& c, float (&v)[4]) {
  if (c.vec) {
    const float4 t = *reinterpret_cast<const float4*>(p + row_off + c.j);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (c.j + e < c.N) ? p[row_off + c.j + e] : 0.0f;
  }
}
__device__ __forceinline__ void st4(float* __restrict__ p, size_t row_off, const ColLane& c, const float (&v)[4]) {
  if (c.vec) {
    *reinterpret_cast<float4*>(p + row_off + c.j) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c.j + e < c.N) p[row_off + c.j + e] = v[e];
  }
}

// sums the two statistic planes of `part` over the row tiles for this lane's 4 columns (fp64, fixed order)
__device__ __forceinline__ void sum_partials(const float* __restrict__ part, int n_tiles, const ColLane& c, int rl,
                                             double (*lds)[32][8], int cl, double (&s1)[4], double (&s2)[4]) {
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = rl; t < n_tiles; t += 8) {
    float v1[4], v2[4];
    ld4(part, (size_t)t * c.N, c, v1);
    ld4(part, ((size_t)n_tiles + t) * c.N, c, v2);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] += (double)v1[e]; a[4 + e] += (double)v2[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) lds[rl][cl][e] = a[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { t1 += lds[k][cl][e]; t2 += lds[k][cl][4 + e]; }
    s1[e] = t1;
    s2[e] = t2;
  }
}

// Synchronized BatchNorm: the row-tile partial sums [2][n_tiles][N] of this rank are reduced to [2][N] IN PLACE before the
// exchange (fp64 accumulation in tile order, one thread per column: it reads all of its 2 n_tiles inputs before it writes the
// two slots, which are inputs of that thread only), so the collective's length is 2 N whatever the rank's row count and tile
// size -- ranks with different shard sizes exchange the same number of floats -- and the apply kernels read
// n_tiles = 1.
__global__ __launch_bounds__(256) void part_reduce_kernel(float* __restrict__ part, int n_tiles, int N) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  double s1 = 0.0, s2 = 0.0;
  for (int t = 0; t < n_tiles; ++t) {
    s1 += (double)part[(size_t)t * N + col];
    s2 += (double)part[((size_t)n_tiles + t) * N + col];
  }
  part[col] = (float)s1;
  part[(size_t)N + col] = (float)s2;
}

struct BnFwdArgs {
  const float* a;  // [B][N] post-ReLU
  float* h;        // [B][N] out
  int B, N, has_bn, n_tiles;
  int Bstat;          // rows the statistics are taken over (= B; synchronized BatchNorm: the global batch)
  const float* part;  // [2][n_tiles][N]: sums of a and a^2 per row tile
  const float* gamma;
  const float* beta;
  float* mmean;
  float* mvar;
  float* save_mean;  // [N], rstd at save_mean + N
  DropArgs drop;
};

// training-mode BatchNorm (batch mean, population variance) + inverted dropout
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(BnFwdArgs p) {
  __shared__ double lds[8][32][8];
  const int tid = threadIdx.x, cl = tid & 31, rl = tid >> 5;
  ColLane c;
  c.j = blockIdx.x * 128 + 4 * cl;
  c.N = p.N;
  c.vec = ((p.N & 3) == 0) && (c.j + 3 < p.N);
  const bool live = c.j < p.N;
  const int step = p.drop.step_dev ? *p.drop.step_dev : 0;
  float mean[4] = {0, 0, 0, 0}, rstd[4] = {1, 1, 1, 1}, ga[4] = {1, 1, 1, 1}, be[4] = {0, 0, 0, 0};
  // this thread's activations start their trip before the statistics are reduced (one memory round trip less)
  float xin[kApplyRows / 8][4];
#pragma unroll
  for (int i = 0; i < kApplyRows / 8; ++i) {
    const int b = blockIdx.y * kApplyRows + rl + 8 * i;
#pragma unroll
    for (int e = 0; e < 4; ++e) xin[i][e] = 0.0f;
    if (live && b < p.B) ld4(p.a, (size_t)b * p.N, c, xin[i]);
  }
  if (p.has_bn) {
    double s1[4], s2[4];
    ColLane cc = c;
    if (!live) { cc.j = 0; cc.vec = false; cc.N = 0; }
    sum_partials(p.part, p.n_tiles, cc, rl, lds, cl, s1, s2);
    if (live) {
      float var[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double m = s1[e] / (double)p.Bstat;
        double v = s2[e] / (double)p.Bstat - m * m;
        v = v > 0.0 ? v : 0.0;
        mean[e] = (float)m;
        var[e] = (float)v;
        rstd[e] = (float)(1.0 / sqrt(v + (double)kBnEps));
      }
      ld4(p.gamma, 0, c, ga);
      ld4(p.beta, 0, c, be);
      if (blockIdx.y == 0 && rl == 0) {
        float mm[4], mv[4];
        ld4(p.mmean, 0, c, mm);
        ld4(p.mvar, 0, c, mv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          mm[e] = mm[e] * kBnMomentum + mean[e] * (1.0f - kBnMomentum);
          mv[e] = mv[e] * kBnMomentum + var[e] * (1.0f - kBnMomentum);
        }
        st4(p.mmean, 0, c, mm);
        st4(p.mvar, 0, c, mv);
        st4(p.save_mean, 0, c, mean);
        st4(p.save_mean, (size_t)p.N, c, rstd);
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < kApplyRows / 8; ++i) {
    const int b = blockIdx.y * kApplyRows + rl + 8 * i;
    if (b >= p.B) break;
    const size_t ro = (size_t)b * p.N;
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = xin[i][e];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (p.has_bn) x[e] = (x[e] - mean[e]) * rstd[e] * ga[e] + be[e];
      x[e] *= dropout_mult(p.drop, step, ro + c.j + e);
    }
    st4(p.h, ro, c, x);
  }
}

struct BnBwdArgs {
  const float* g;   // [B][N] dh * dropout (from the producing GEMM's epilogue)
  const float* a;   // [B][N] post-ReLU
  float* dz;        // [B][N] gradient w.r.t. the pre-activation
  int B, N, n_tiles;
  int Bstat;          // rows the statistics were taken over (= B; synchronized BatchNorm: the global batch)
  float grad_scale;   // 1, or 1 / world when the partial sums were all-reduced (the gradient all-reduce sums them again)
  const float* part;  // [2][n_tiles][N]: sums of g and g * xhat per row tile
  const float* gamma;
  const float* save_mean;
  float* dgamma;
  float* dbeta;
  unsigned* amax_out;  // arithmetic mode 2: max |dz| is folded into this word (see GemmArgs::amax_out); may be null
};

// backward of BatchNorm(train) -> ReLU given the column sums; also writes dgamma / dbeta
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(BnBwdArgs p) {
  __shared__ double lds[8][32][8];
  const int tid = threadIdx.x, cl = tid & 31, rl = tid >> 5;
  ColLane c;
  c.j = blockIdx.x * 128 + 4 * cl;
  c.N = p.N;
  c.vec = ((p.N & 3) == 0) && (c.j + 3 < p.N);
  const bool live = c.j < p.N;
  float gin[kApplyRows / 8][4], ain[kApplyRows / 8][4];
#pragma unroll
  for (int i = 0; i < kApplyRows / 8; ++i) {
    const int b = blockIdx.y * kApplyRows + rl + 8 * i;
#pragma unroll
    for (int e = 0; e < 4; ++e) { gin[i][e] = 0.0f; ain[i][e] = 0.0f; }
    if (live && b < p.B) {
      ld4(p.g, (size_t)b * p.N, c, gin[i]);
      ld4(p.a, (size_t)b * p.N, c, ain[i]);
    }
  }
  double s1[4], s2[4];
  ColLane cc = c;
  if (!live) { cc.j = 0; cc.vec = false; cc.N = 0; }
  sum_partials(p.part, p.n_tiles, cc, rl, lds, cl, s1, s2);
  float omax = 0.0f;
  if (live) {
  float mean[4], rstd[4], ga[4], dbt[4], dg[4];
  ld4(p.save_mean, 0, c, mean);
  ld4(p.save_mean, (size_t)p.N, c, rstd);
  ld4(p.gamma, 0, c, ga);
#pragma unroll
  for (int e = 0; e < 4; ++e) { dbt[e] = (float)s1[e]; dg[e] = (float)s2[e]; }
  if (blockIdx.y == 0 && rl == 0) {
    float ob[4], og[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { ob[e] = dbt[e] * p.grad_scale; og[e] = dg[e] * p.grad_scale; }
    st4(p.dbeta, 0, c, ob);
    st4(p.dgamma, 0, c, og);
  }
  const float invB = 1.0f / (float)p.Bstat;
#pragma unroll
  for (int i = 0; i < kApplyRows / 8; ++i) {
    const int b = blockIdx.y * kApplyRows + rl + 8 * i;
    if (b >= p.B) break;
    const size_t ro = (size_t)b * p.N;
    float gv[4], av[4], o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { gv[e] = gin[i][e]; av[e] = ain[i][e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (av[e] - mean[e]) * rstd[e];
      const float d = ga[e] * rstd[e] * (gv[e] - dbt[e] * invB - xh * dg[e] * invB);
      o[e] = av[e] > 0.0f ? d : 0.0f;
      if (c.j + e < c.N) omax = fmaxf(omax, fabsf(o[e]));
    }
    st4(p.dz, ro, c, o);
  }
  }
  if (p.amax_out) amax_publish(p.amax_out, omax);
}

// upstream vector at the network output -> dz at the logits, one thread per row (class_gradient and the
// optimisation-based attacks): on_logits: dz = v;  else out = softmax(z): dz = p * (v - sum_c p_c v_c)
__global__ __launch_bounds__(256) void softmax_vjp_kernel(const float* __restrict__ z, const float* __restrict__ v, int B,
                                                           int C, int on_logits, float* __restrict__ prob,
                                                           float* __restrict__ dz) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* zr = z + (size_t)b * C;
  const float* vr = v + (size_t)b * C;
  float mx = zr[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, zr[c]);
  float se = 0.0f;
  for (int c = 0; c < C; ++c) se += expf(zr[c] - mx);
  const float inv = 1.0f / se;
  float dot = 0.0f;
  for (int c = 0; c < C; ++c) dot = fmaf(expf(zr[c] - mx) * inv, vr[c], dot);
  for (int c = 0; c < C; ++c) {
    const float pc = expf(zr[c] - mx) * inv;
    if (prob) prob[(size_t)b * C + c] = pc;
    dz[(size_t)b * C + c] = on_logits ? vr[c] : pc * (vr[c] - dot);
  }
}

// softmax + categorical cross-entropy from logits, one thread per row.
//   prob (optional), dz = (p - y) * inv_batch (optional), loss_rows = -sum y log_softmax(z) (optional),
//   correct_rows = [argmax p == argmax y] (optional), onehot_out = one-hot argmax z (optional)
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ z, const float* __restrict__ y, int B,
                                                          int C, float inv_batch, float* __restrict__ prob,
                                                          float* __restrict__ dz, float* __restrict__ loss_rows,
                                                          float* __restrict__ correct_rows,
                                                          float* __restrict__ onehot_out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* zr = z + (size_t)b * C;
  float mx = zr[0];
  int am = 0;
  for (int c = 1; c < C; ++c)
    if (zr[c] > mx) { mx = zr[c]; am = c; }
  float se = 0.0f;
  for (int c = 0; c < C; ++c) se += expf(zr[c] - mx);
  const float lse = logf(se);
  const float inv = 1.0f / se;
  float loss = 0.0f, ymax = -INFINITY;
  int ay = 0;
  for (int c = 0; c < C; ++c) {
    const float zs = zr[c] - mx;
    const float pc = expf(zs) * inv;
    if (prob) prob[(size_t)b * C + c] = pc;
    if (y) {
      const float yc = y[(size_t)b * C + c];
      if (yc != 0.0f) loss -= yc * (zs - lse);
      if (yc > ymax) { ymax = yc; ay = c; }
      if (dz) dz[(size_t)b * C + c] = (pc - yc) * inv_batch;
    }
    if (onehot_out) onehot_out[(size_t)b * C + c] = (c == am) ? 1.0f : 0.0f;
  }
  if (loss_rows) loss_rows[b] = loss;
  if (correct_rows) correct_rows[b] = (am == ay) ? 1.0f : 0.0f;
}

static inline size_t align4(size_t x) { return (x + 3) & ~size_t(3); }

void mlp_plan_free(lipasr_mlp* m) {
  if (!m) return;
  if (m->ws) (void)hipFree(m->ws);
  if (m->xc_gran) (void)hipFree(m->xc_gran);
  if (m->xc_ctrl) (void)hipFree(m->xc_ctrl);
  delete m;
}

}  // namespace lipasr

using namespace lipasr;

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
extern "C" {

long lipasr_debug_launch_count(int kind) { return (kind == 0 || kind == 1) ? g_launch_count[kind] : -1; }

int lipasr_debug_gemm_mode(int mode) {
  g_gemm_mode = mode & 3;  // (bits: 0-1 kernel choice, 2 split dW_0, 3 grouped launch on fragment tiles, 4 XCD-aware tile map, 5 no LDS-DMA ring, 6 64 x 64 ring tile for the weight gradients, 7 the 128 x 128 tile without the split pass, 8 no 128 x 64 exchange tiles, 9 no loader instance of the 64 x 64 exchange ring tile)
  g_split_dw0 = (mode >> 2) & 1;
  g_group_lds = ((mode >> 3) & 1) ? 0 : 1;
  g_xcd_map = (mode >> 4) & 1;
  g_no_ring = (mode >> 5) & 1;
  g_ring_tile = ((mode >> 6) & 1) ? 1 : ((mode >> 7) & 1) ? 3 : 2;
  g_ring2 = ((mode >> 8) & 1) ? 0 : 1;
  g_ring_x1 = ((mode >> 9) & 1) ? 0 : 1;
  return LIPASR_OK;
}

int lipasr_gemm_f32(lipasr_handle_t h, int transA, int transB, int M, int N, int K, const float* A, int lda,
                    const float* B, int ldb, float* C, int ldc, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && A && B && C, "lipasr_gemm_f32: null argument");
  LP_CHECK_ARG(M > 0 && N > 0 && K > 0, "lipasr_gemm_f32: empty problem %dx%dx%d", M, N, K);
  LP_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "lipasr_gemm_f32: leading dimension too small");
  GemmArgs g = gemm_args(A, lda, B, ldb, C, ldc, M, N, K, EPI_STORE);
  return launch_gemm(transA ? 1 : 0, transB ? 0 : 1, g, S(stream));
}

int lipasr_gemm_f16x2(lipasr_handle_t h, int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B,
                      int ldb, float* C, int ldc, float scale_a, float scale_b, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && A && B && C, "lipasr_gemm_f16x2: null argument");
  LP_CHECK_ARG(M > 0 && N > 0 && K > 0, "lipasr_gemm_f16x2: empty problem %dx%dx%d", M, N, K);
  LP_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "lipasr_gemm_f16x2: leading dimension too small");
  int ea = 0, eb = 0;
  LP_CHECK_ARG(scale_a > 0.0f && scale_b > 0.0f && frexpf(scale_a, &ea) == 0.5f && frexpf(scale_b, &eb) == 0.5f,
               "lipasr_gemm_f16x2: the scales must be powers of two (got %g, %g)", (double)scale_a, (double)scale_b);
  GemmArgs g = gemm_args(A, lda, B, ldb, C, ldc, M, N, K, EPI_STORE);
  g.bf16 = 2;
  g.sa = scale_a;
  g.sb = scale_b;
  g.zeros = h->zeros;
  return launch_gemm(transA ? 1 : 0, transB ? 0 : 1, g, S(stream));
}

int lipasr_mlp_create(lipasr_handle_t h, int n_layers, const int* widths, const int* bn, const float* dropout,
                      const int* nonneg, int max_batch, lipasr_mlp_t* out) {
  LP_CHECK_ARG(h && widths && out, "lipasr_mlp_create: null argument");
  LP_CHECK_ARG(n_layers >= 1 && n_layers <= LIPASR_MAX_LAYERS, "lipasr_mlp_create: n_layers=%d outside [1,%d]", n_layers,
               LIPASR_MAX_LAYERS);
  LP_CHECK_ARG(max_batch >= 1, "lipasr_mlp_create: max_batch=%d", max_batch);
  for (int l = 0; l <= n_layers; ++l) LP_CHECK_ARG(widths[l] >= 1, "lipasr_mlp_create: widths[%d]=%d", l, widths[l]);
  LP_CHECK_ARG(widths[n_layers] <= 32, "lipasr_mlp_create: %d classes; at most 32 are supported", widths[n_layers]);
  lipasr_mlp* m = new lipasr_mlp();
  m->ctx = h;
  m->n_layers = n_layers;
  m->max_batch = max_batch;
  size_t po = 0, so = 0, wo = 0;
  int maxw = widths[0];
  for (int l = 0; l < n_layers; ++l) {
    MlpLayer& L = m->L[l];
    L.n_in = widths[l];
    L.n_out = widths[l + 1];
    const bool last = (l == n_layers - 1);
    L.bn = !last && bn && bn[l];
    L.dropout = (!last && dropout) ? dropout[l] : 0.0f;
    L.nonneg = nonneg && nonneg[l];
    if (L.dropout < 0.0f || L.dropout >= 1.0f) {
      delete m;
      set_error("lipasr_mlp_create: dropout[%d]=%g outside [0,1)", l, (double)L.dropout);
      return LIPASR_EINVAL;
    }
    maxw = L.n_out > maxw ? L.n_out : maxw;
    L.offW = po; po = align4(po + (size_t)L.n_in * L.n_out);
    L.offb = po; po = align4(po + L.n_out);
    if (L.bn) {
      L.offg = po; po = align4(po + L.n_out);
      L.offbe = po; po = align4(po + L.n_out);
      L.offmm = so; so = align4(so + L.n_out);
      L.offmv = so; so = align4(so + L.n_out);
    }
    if (!last) {
      L.offA = wo; wo = align4(wo + (size_t)max_batch * L.n_out);
      if (L.bn || L.dropout > 0.0f) { L.offH = wo; wo = align4(wo + (size_t)max_batch * L.n_out); }
      else L.offH = L.offA;
      L.offMean = wo; wo = align4(wo + 2 * (size_t)L.n_out);
    }
    L.offDz = wo; wo = align4(wo + (size_t)max_batch * L.n_out);
  }
  m->n_params = po;
  m->n_state = so;
  m->max_width = maxw;
  const size_t C = widths[n_layers];
  m->offLogits = wo; wo = align4(wo + (size_t)max_batch * C);
  m->offProb = wo; wo = align4(wo + (size_t)max_batch * C);
  m->offDzLast = wo; wo = align4(wo + (size_t)max_batch * C);
  m->offG0 = wo; wo = align4(wo + (size_t)max_batch * maxw);
  m->offG1 = wo; wo = align4(wo + (size_t)max_batch * maxw);
  m->offG2 = wo; wo = align4(wo + (size_t)max_batch * maxw);
  m->offPart = wo; wo = align4(wo + 2 * (size_t)((max_batch + 31) / 32) * maxw);  // column partials of the *_STATS epilogues
  m->ws_floats = wo;
  DeviceGuard g(h->device);
  if (hipMalloc(&m->ws, wo * sizeof(float)) != hipSuccess) {
    delete m;
    set_error("lipasr_mlp_create: workspace allocation of %zu bytes failed", wo * sizeof(float));
    return LIPASR_ENOMEM;
  }
  (void)hipMemset(m->ws, 0, wo * sizeof(float));
  // exchange epilogue state (round 5): granules and control words per BatchNorm layer and direction, all zero (tag 0 is never used)
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) m->n_cus = prop.multiProcessorCount;
    m->xc_rt_max = std::min((max_batch + 31) / 32, 64);
    const size_t amax_words = (size_t)LIPASR_MAX_LAYERS * kAmaxSlots * kAmaxStride;
    size_t go = 0, co = 16 + amax_words;  // control words 0 .. 15: the error word (its own 64-byte slot), then amax[layer][slot]
    for (int dir = 0; dir < 2; ++dir)
      for (int l = 0; l + 1 < n_layers; ++l) {
        if (!m->L[l].bn) continue;
        const size_t nblk = (size_t)(m->L[l].n_out + 31) / 32;
        m->xc_gran_off[dir][l] = go; go += nblk * m->xc_rt_max * 128;
        m->xc_ctrl_off[dir][l] = co; co += nblk * 32;
      }
    if (hipMalloc(&m->xc_ctrl, co * sizeof(unsigned)) != hipSuccess) {
      (void)hipGetLastError();
      mlp_plan_free(m);
      set_error("lipasr_mlp_create: control-word allocation failed");
      return LIPASR_ENOMEM;
    }
    (void)hipMemset(m->xc_ctrl, 0, co * sizeof(unsigned));
    m->xc_err = reinterpret_cast<int*>(m->xc_ctrl);
    m->amax = m->xc_ctrl + 16;  // layer l: m->amax + l * kAmaxSlots * kAmaxStride
    if (go > 0) {
      if (hipMalloc(&m->xc_gran, go * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();
        m->xc_gran = nullptr;  // no exchange state: the launch chain is used
      } else {
        (void)hipMemset(m->xc_gran, 0, go * sizeof(unsigned long long));
      }
    }
  }
  h->mlps.push_back(m);
  *out = m;
  return LIPASR_OK;
}

int lipasr_mlp_destroy(lipasr_mlp_t m) {
  LP_CHECK_ARG(m != nullptr, "lipasr_mlp_destroy: null plan");
  DeviceGuard g(m->ctx->device);
  std::vector<lipasr_mlp*>& v = m->ctx->mlps;
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == m) { v.erase(v.begin() + i); break; }
  lipasr::mlp_plan_free(m);
  return LIPASR_OK;
}

int lipasr_mlp_sizes(lipasr_mlp_t m, size_t* n_params, size_t* n_state) {
  LP_CHECK_ARG(m && n_params && n_state, "lipasr_mlp_sizes: null argument");
  *n_params = m->n_params;
  *n_state = m->n_state;
  return LIPASR_OK;
}

int lipasr_mlp_segment(lipasr_mlp_t m, int layer, int kind, size_t* offset, size_t* count) {
  LP_CHECK_ARG(m && offset && count, "lipasr_mlp_segment: null argument");
  LP_CHECK_ARG(layer >= 0 && layer < m->n_layers, "lipasr_mlp_segment: layer %d out of range", layer);
  const MlpLayer& L = m->L[layer];
  switch (kind) {
    case LIPASR_SEG_W: *offset = L.offW; *count = (size_t)L.n_in * L.n_out; break;
    case LIPASR_SEG_B: *offset = L.offb; *count = L.n_out; break;
    case LIPASR_SEG_GAMMA: *offset = L.offg; *count = L.bn ? L.n_out : 0; break;
    case LIPASR_SEG_BETA: *offset = L.offbe; *count = L.bn ? L.n_out : 0; break;
    case LIPASR_SEG_MMEAN: *offset = L.offmm; *count = L.bn ? L.n_out : 0; break;
    case LIPASR_SEG_MVAR: *offset = L.offmv; *count = L.bn ? L.n_out : 0; break;
    default: set_error("lipasr_mlp_segment: unknown kind %d", kind); return LIPASR_EINVAL;
  }
  return LIPASR_OK;
}

}  // extern "C"

namespace lipasr {

static int check_batch(const char* fn, lipasr_mlp_t m, int batch) {
  LP_CHECK_ARG(m != nullptr, "%s: null plan", fn);
  LP_CHECK_ARG(batch >= 1 && batch <= m->max_batch, "%s: batch %d outside [1, %d]", fn, batch, m->max_batch);
  return LIPASR_OK;
}

static DropArgs drop_for_layer(const lipasr_mlp* m, int l, const lipasr_dropout_cfg* cfg) {
  DropArgs d;
  memset(&d, 0, sizeof(d));
  d.rate = m->L[l].dropout;
  d.layer = l;
  if (!cfg || cfg->mode == 0 || d.rate <= 0.0f) { d.mode = 0; return d; }
  d.mode = cfg->mode;
  d.seed = cfg->seed;
  d.step_dev = cfg->step_dev;
  d.mask = (cfg->mode == 2 && cfg->masks) ? cfg->masks[l] : nullptr;
  if (cfg->mode == 2 && d.mask == nullptr) d.mode = 0;
  return d;
}

// inference-mode forward: fills A_l (post-ReLU, if keep_a) and H_l, logits into m->ws
// ce_y != nullptr (and <= 32 classes): the last GEMM's epilogue also produces softmax, and (p - y) / batch at
// m->ws + offDzLast -- the caller then skips softmax_ce_kernel.  Returns through *fused whether it did.
static int forward_infer(lipasr_mlp* m, const float* params, const float* bnstate, const float* x, int batch,
                         bool keep_a, float* logits_out, hipStream_t st, const float* ce_y = nullptr,
                         bool* fused = nullptr) {
  const float* hin = x;
  if (fused) *fused = false;
  for (int l = 0; l < m->n_layers; ++l) {
    const MlpLayer& L = m->L[l];
    const bool last = (l == m->n_layers - 1);
    float* outp = last ? logits_out : (m->ws + L.offH);
    const bool fuse = last && ce_y && L.n_out <= 32;
    GemmArgs g = gemm_args(hin, L.n_in, params + L.offW, L.n_out, outp, L.n_out, batch, L.n_out, L.n_in,
                           last ? (fuse ? EPI_BIAS_SOFTMAX_CE : EPI_BIAS) : EPI_BIAS_RELU_BN);
    g.bias = params + L.offb;
    if (fuse) {
      g.y = ce_y;
      g.inv_batch = 1.0f / (float)batch;
      g.prob = m->ws + m->offProb;
      g.dz = m->ws + m->offDzLast;
      if (fused) *fused = true;
    }
    if (!last) {
      if (L.bn) {
        g.gamma = params + L.offg;
        g.beta = params + L.offbe;
        g.mmean = bnstate + L.offmm;
        g.mvar = bnstate + L.offmv;
      }
      // when H aliases A (no BN, no dropout) the ReLU output lands in H == A directly
      g.aux = (keep_a && L.offH != L.offA) ? (m->ws + L.offA) : nullptr;
    }
    set_arith(g, m, OP_ACT, OP_WEIGHT, 1.0f, false);
    int rc = launch_gemm(0, 1, g, st);
    if (rc != LIPASR_OK) return rc;
    hin = outp;
  }
  return LIPASR_OK;
}

// inference-mode backward to the input from dz at the logits (in m->ws + offDzLast).
// final_mode 0: store dx; 1: fused sign step on x_adv.
static int backward_infer(lipasr_mlp* m, const float* params, const float* bnstate, int batch, float* dx, float* x_adv,
                          const float* x0, float alpha, float eps, hipStream_t st, float g0 = 1.0f) {
  const float* gin = m->ws + m->offDzLast;
  float* pp[2] = {m->ws + m->offG0, m->ws + m->offG1};
  int cur = 0;
  for (int l = m->n_layers - 1; l >= 0; --l) {
    const MlpLayer& L = m->L[l];
    // dprev[B][n_in] = gin[B][n_out] * W^T ; W stored [n_in][n_out] -> K (= n_out) contiguous
    if (l > 0) {
      const MlpLayer& P = m->L[l - 1];
      GemmArgs g = gemm_args(gin, L.n_out, params + L.offW, L.n_out, pp[cur], L.n_in, batch, L.n_in, L.n_out,
                             EPI_DZ_INFER);
      if (P.bn) {
        g.gamma = params + P.offg;
        g.mvar = bnstate + P.offmv;
      }
      g.aux = m->ws + P.offA;
      set_arith(g, m, OP_GRAD, OP_WEIGHT, g0, false);
      int rc = launch_gemm(0, 0, g, st);
      if (rc != LIPASR_OK) return rc;
      gin = pp[cur];
      cur ^= 1;
    } else {
      GemmArgs g = gemm_args(gin, L.n_out, params + L.offW, L.n_out, dx, L.n_in, batch, L.n_in, L.n_out,
                             x_adv ? EPI_SIGNSTEP : EPI_STORE);
      g.x_adv = x_adv;
      g.x0 = x0;
      g.alpha = alpha;
      g.eps = eps;
      set_arith(g, m, OP_GRAD, OP_WEIGHT, g0, false);
      int rc = launch_gemm(0, 0, g, st);
      if (rc != LIPASR_OK) return rc;
    }
  }
  return LIPASR_OK;
}

}  // namespace lipasr

extern "C" {

// dw_mode 0: every weight gradient from ONE grouped launch; 1: the first layer's [dW; db] as its own launch after the
// grouped launch of the others (LDS-tiled 64x64 kernel when legal); 2: the first layer's left out (lipasr_mlp_train_dw0)
// Segments (synchronized BatchNorm under data parallelism): the launch sequence is cut after every GEMM whose epilogue
// leaves BatchNorm column partial sums (forward: sums of a, a^2; backward: sums of g, g xhat), i.e. right before the
// apply kernel that consumes them.  seg < 0 runs everything; otherwise only the launches of segment `seg`, and the caller
// SUM-all-reduces the partials (segment_exchange_floats) before it runs the next one.  part_ext: caller-owned partials
// buffer (a torch tensor the caller can hand to its collective), stat_batch: rows the statistics cover (global batch).
struct SegArgs {
  int seg = -1;
  float* part_ext = nullptr;
  int stat_batch = 0;
  float grad_scale = 1.0f;
};

static int train_fwd_bwd_impl(lipasr_mlp_t m, const float* params, float* bnstate, const float* x, const float* y_onehot,
                              int batch, float inv_batch, const lipasr_dropout_cfg* dropout, float* grads,
                              float* loss_rows, float* correct_rows, float* probs, lipasr_stream_t stream, int dw_mode,
                              const SegArgs& sa = SegArgs()) {
  int rc = check_batch("lipasr_mlp_train_fwd_bwd", m, batch);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(params && x && y_onehot && grads, "lipasr_mlp_train_fwd_bwd: null argument");
  LP_CHECK_ARG(m->n_state == 0 || bnstate, "lipasr_mlp_train_fwd_bwd: bnstate is null");
  LP_CHECK_ARG(!dropout || (dropout->mode >= 0 && dropout->mode <= 2), "lipasr_mlp_train_fwd_bwd: dropout mode %d",
               dropout ? dropout->mode : 0);
  hipStream_t st = S(stream);
  m->last_inv_batch = inv_batch;
  const int Lc = m->n_layers;
  const int C = m->L[Lc - 1].n_out;
  float* ws = m->ws;
  float* part = sa.part_ext ? sa.part_ext : (ws + m->offPart);
  const dim3 apply_block(256);
  const int bstat = sa.stat_batch > 0 ? sa.stat_batch : batch;
  int cur = 0;  // current segment
#define LP_ON (sa.seg < 0 || sa.seg == cur)

  // ---- forward (training mode): GEMM (+bias, ReLU, column partials) -> BatchNorm/dropout apply
  const float* hin = x;
  for (int l = 0; l < Lc; ++l) {
    const MlpLayer& L = m->L[l];
    const bool last = (l == Lc - 1);
    float* outp = last ? (ws + m->offLogits) : (ws + L.offA);
    const bool fuse_ce = last && C <= 32;  // softmax, loss and (p - y) / B in the last GEMM's epilogue
    // BatchNorm inside this GEMM (exchange epilogue) where the whole grid can be resident; not with synchronized BatchNorm,
    // whose sums leave the device between the GEMM and the apply kernel
    const bool bnx = !last && L.bn && sa.seg < 0 && bnx_fits(m, true, batch, L.n_out, L.n_in);
    GemmArgs g = gemm_args(hin, L.n_in, params + L.offW, L.n_out, outp, L.n_out, batch, L.n_out, L.n_in,
                           last ? (fuse_ce ? EPI_BIAS_SOFTMAX_CE : EPI_BIAS) : (L.bn ? (bnx ? EPI_BIAS_RELU_BNX : EPI_BIAS_RELU_STATS) : EPI_BIAS_RELU));
    g.bias = params + L.offb;
    g.part = part;
    if (m->compute_bf16 == 2 && !last) g.amax_zero = m->amax + (size_t)l * kAmaxSlots * kAmaxStride;  // the backward pass of this step folds max |dz_l| into it
    if (bnx) {
      g.xc_gran = m->xc_gran + m->xc_gran_off[0][l]; g.xc_ctrl = m->xc_ctrl + m->xc_ctrl_off[0][l]; g.xc_err = m->xc_err; g.xc_rt_max = m->xc_rt_max;
      // LIPASR_XC_NOWAIT=1 (timing probe only, results are WRONG): the exchange epilogue without its wait and sweep, to see what the
      // exchange instances cost apart from the exchange
      static const bool xc_nowait = getenv("LIPASR_XC_NOWAIT") != nullptr;
      g.Bstat = xc_nowait ? -bstat : bstat;
      g.h_out = ws + L.offH;
      g.gamma = params + L.offg; g.beta = params + L.offbe;
      g.mmean_w = bnstate + L.offmm; g.mvar_w = bnstate + L.offmv; g.save_w = ws + L.offMean;
      g.drop = drop_for_layer(m, l, dropout);
    }
    if (fuse_ce) {
      g.y = y_onehot;
      g.inv_batch = inv_batch;
      g.prob = probs ? probs : (ws + m->offProb);
      g.dz = ws + m->offDzLast;
      g.loss_rows = loss_rows;
      g.correct_rows = correct_rows;
    }
    set_arith(g, m, OP_ACT, OP_WEIGHT);
    g.lds_min_tiles = m->lds_min_tiles;
    if (LP_ON) {
      rc = launch_gemm(0, 1, g, st);
      if (rc != LIPASR_OK) return rc;
      if (sa.seg >= 0 && !last && L.bn) {  // synchronized BatchNorm: [2][row tiles][N] -> [2][N] before the exchange
        hipLaunchKernelGGL(part_reduce_kernel, dim3((L.n_out + 255) / 256), dim3(256), 0, st, part,
                           stats_row_tiles(batch, L.n_out, L.n_in, m->lds_min_tiles), L.n_out);
        LP_LAUNCH_CHECK();
      }
    }
    if (!last && L.bn) ++cur;  // exchange point: the partial sums of a, a^2 are complete
    if (!last && L.offH != L.offA && LP_ON && !bnx) {
      BnFwdArgs b;
      memset(&b, 0, sizeof(b));
      b.a = ws + L.offA; b.h = ws + L.offH; b.B = batch; b.N = L.n_out; b.has_bn = L.bn ? 1 : 0;
      b.Bstat = bstat;
      b.part = part;
      b.n_tiles = sa.seg >= 0 ? 1 : stats_row_tiles(batch, L.n_out, L.n_in, m->lds_min_tiles);
      if (L.bn) {
        b.gamma = params + L.offg; b.beta = params + L.offbe;
        b.mmean = bnstate + L.offmm; b.mvar = bnstate + L.offmv;
        b.save_mean = ws + L.offMean;
      }
      b.drop = drop_for_layer(m, l, dropout);
      const dim3 grid((L.n_out + 127) / 128, (batch + kApplyRows - 1) / kApplyRows);
      hipLaunchKernelGGL(bn_apply_fwd_kernel, grid, apply_block, 0, st, b);
      LP_LAUNCH_CHECK();
    }
    hin = (!last && L.offH != L.offA) ? (ws + L.offH) : outp;
  }
  // ---- loss and gradient at the logits (already done by the last GEMM's epilogue for <= 32 classes)
  if (C > 32 && LP_ON) {
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, ws + m->offLogits, y_onehot, batch,
                       C, inv_batch, probs ? probs : (ws + m->offProb), ws + m->offDzLast, loss_rows, correct_rows,
                       (float*)nullptr);
    LP_LAUNCH_CHECK();
  }

  // ---- backward.  The dX chain runs first, layer by layer (dX GEMM fused with the dropout backward and the
  // BatchNorm column sums, then the BatchNorm/ReLU backward apply), keeping every layer's pre-activation gradient;
  // all weight gradients then come from ONE grouped launch.
  float* tmp = ws + m->offG0;
  for (int l = Lc - 1; l >= 1; --l) {
    const MlpLayer& L = m->L[l];
    const MlpLayer& P = m->L[l - 1];
    const float* gin = (l == Lc - 1) ? (ws + m->offDzLast) : (ws + L.offDz);
    // dh_prev[B][n_in] = gin[B][n_out] * W^T
    const bool bnx = P.bn && sa.seg < 0 && bnx_fits(m, false, batch, L.n_in, L.n_out);
    float* out1 = (P.bn && !bnx) ? tmp : (ws + P.offDz);
    GemmArgs gx = gemm_args(gin, L.n_out, params + L.offW, L.n_out, out1, L.n_in, batch, L.n_in, L.n_out,
                            P.bn ? (bnx ? EPI_DH_BNX : EPI_DH_STATS) : EPI_DZ_NOBN);
    gx.aux = ws + P.offA;
    gx.drop = drop_for_layer(m, l - 1, dropout);
    gx.part = part;
    if (P.bn) gx.save_mean = ws + P.offMean;
    if (bnx) {
      gx.xc_gran = m->xc_gran + m->xc_gran_off[1][l - 1]; gx.xc_ctrl = m->xc_ctrl + m->xc_ctrl_off[1][l - 1]; gx.xc_err = m->xc_err;
      gx.xc_rt_max = m->xc_rt_max;
      gx.Bstat = getenv("LIPASR_XC_NOWAIT") ? -bstat : bstat; gx.grad_scale = sa.grad_scale;
      gx.gamma = params + P.offg;
      gx.dgamma = grads + P.offg; gx.dbeta = grads + P.offbe;
    }
    set_arith(gx, m, OP_GRAD, OP_WEIGHT, inv_batch);
    // arithmetic mode 2: the size of a gradient is not known beforehand (BatchNorm's rstd can amplify it 30-fold per layer in an
    // untrained network): whoever writes dz_l folds max |dz_l| into amax[l], whoever multiplies with dz_l derives its scale from it
    const bool dyn = m->compute_bf16 == 2;
    if (dyn && l < Lc - 1 && L.bn) gx.sa_dyn = m->amax + (size_t)l * kAmaxSlots * kAmaxStride;
    if (dyn && P.bn) gx.amax_out = m->amax + (size_t)(l - 1) * kAmaxSlots * kAmaxStride;
    gx.lds_min_tiles = m->lds_min_tiles;
    if (LP_ON) {
      rc = launch_gemm(0, 0, gx, st);
      if (rc != LIPASR_OK) return rc;
      if (sa.seg >= 0 && P.bn) {
        hipLaunchKernelGGL(part_reduce_kernel, dim3((P.n_out + 255) / 256), dim3(256), 0, st, part,
                           stats_row_tiles(batch, P.n_out, L.n_out, m->lds_min_tiles), P.n_out);
        LP_LAUNCH_CHECK();
      }
    }
    if (P.bn) ++cur;  // exchange point: the partial sums of g, g xhat are complete
    if (P.bn && LP_ON && !bnx) {
      BnBwdArgs b;
      memset(&b, 0, sizeof(b));
      b.g = tmp; b.a = ws + P.offA; b.dz = ws + P.offDz; b.B = batch; b.N = P.n_out;
      b.Bstat = bstat; b.grad_scale = sa.grad_scale;
      b.part = part;
      b.n_tiles = sa.seg >= 0 ? 1 : stats_row_tiles(batch, P.n_out, L.n_out, m->lds_min_tiles);
      b.gamma = params + P.offg; b.save_mean = ws + P.offMean;
      b.dgamma = grads + P.offg; b.dbeta = grads + P.offbe;
      b.amax_out = dyn ? m->amax + (size_t)(l - 1) * kAmaxSlots * kAmaxStride : nullptr;
      const dim3 grid((P.n_out + 127) / 128, (batch + kApplyRows - 1) / kApplyRows);
      hipLaunchKernelGGL(bn_apply_bwd_kernel, grid, apply_block, 0, st, b);
      LP_LAUNCH_CHECK();
    }
  }
  // [dW ; db] = [lin ; 1]^T[n_in+1][B] * gin[B][n_out] for every layer: the all-ones row yields the bias gradient
  GemmArgs gw[LIPASR_MAX_LAYERS];
  for (int l = 0; l < Lc; ++l) {
    const MlpLayer& L = m->L[l];
    const float* lin = (l == 0) ? x : (ws + m->L[l - 1].offH);
    const float* gin = (l == Lc - 1) ? (ws + m->offDzLast) : (ws + L.offDz);
    gw[l] = gemm_args(lin, L.n_in, gin, L.n_out, grads + L.offW, L.n_out, L.n_in + 1, L.n_out, batch, EPI_STORE);
    gw[l].ones_row = 1;
    gw[l].extra_out = grads + L.offb;
    set_arith(gw[l], m, OP_ACT, OP_GRAD, inv_batch);
    if (m->compute_bf16 == 2 && l < Lc - 1 && L.bn) gw[l].sb_dyn = m->amax + (size_t)l * kAmaxSlots * kAmaxStride;
  }
  if (!LP_ON) return LIPASR_OK;
#undef LP_ON
  if (dw_mode == 0 || Lc == 1) return launch_gemm_group_tn(gw, Lc, st);
  rc = launch_gemm_group_tn(gw + 1, Lc - 1, st);
  if (rc != LIPASR_OK || dw_mode == 2) return rc;
  return launch_gemm(1, 1, gw[0], st);
}

int lipasr_mlp_train_fwd_bwd(lipasr_mlp_t m, const float* params, float* bnstate, const float* x, const float* y_onehot,
                             int batch, float inv_batch, const lipasr_dropout_cfg* dropout, float* grads,
                             float* loss_rows, float* correct_rows, float* probs, lipasr_stream_t stream) {
  return train_fwd_bwd_impl(m, params, bnstate, x, y_onehot, batch, inv_batch, dropout, grads, loss_rows, correct_rows, probs,
                            stream, g_split_dw0 ? 1 : 0);
}

int lipasr_mlp_train_fwd_bwd_head(lipasr_mlp_t m, const float* params, float* bnstate, const float* x, const float* y_onehot,
                                  int batch, float inv_batch, const lipasr_dropout_cfg* dropout, float* grads,
                                  float* loss_rows, float* correct_rows, float* probs, lipasr_stream_t stream) {
  return train_fwd_bwd_impl(m, params, bnstate, x, y_onehot, batch, inv_batch, dropout, grads, loss_rows, correct_rows, probs,
                            stream, 2);
}

int lipasr_mlp_train_dw0(lipasr_mlp_t m, const float* x, int batch, float* grads, lipasr_stream_t stream) {
  int rc = check_batch("lipasr_mlp_train_dw0", m, batch);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(x && grads, "lipasr_mlp_train_dw0: null argument");
  if (m->n_layers == 1) return LIPASR_OK;  // a one-layer plan's head already holds every gradient
  const MlpLayer& L = m->L[0];
  GemmArgs g = gemm_args(x, L.n_in, m->ws + L.offDz, L.n_out, grads + L.offW, L.n_out, L.n_in + 1, L.n_out, batch, EPI_STORE);
  g.ones_row = 1;
  g.extra_out = grads + L.offb;
  set_arith(g, m, OP_ACT, OP_GRAD, m->last_inv_batch);
  if (m->compute_bf16 == 2 && L.bn) g.sb_dyn = m->amax;
  return launch_gemm(1, 1, g, S(stream));
}

int lipasr_mlp_train_segments(lipasr_mlp_t m, int* n_segments) {
  LP_CHECK_ARG(m && n_segments, "lipasr_mlp_train_segments: null argument");
  int n = 1;
  for (int l = 0; l + 1 < m->n_layers; ++l)
    if (m->L[l].bn) n += 2;  // one exchange in the forward pass, one in the backward pass
  *n_segments = n;
  return LIPASR_OK;
}

// floats of the partials buffer to SUM-all-reduce after segment `seg` (0 after the last one)
int lipasr_mlp_train_segment_exchange(lipasr_mlp_t m, int batch, int seg, size_t* floats) {
  int rc = check_batch("lipasr_mlp_train_segment_exchange", m, batch);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(floats != nullptr && seg >= 0, "lipasr_mlp_train_segment_exchange: bad argument");
  *floats = 0;
  int cur = 0;
  for (int l = 0; l + 1 < m->n_layers; ++l)  // forward: layer l's GEMM closes a segment if layer l has BatchNorm
    if (m->L[l].bn) {
      if (cur == seg) { *floats = 2 * (size_t)m->L[l].n_out; return LIPASR_OK; }  // reduced to [2][N] by part_reduce_kernel
      ++cur;
    }
  for (int l = m->n_layers - 1; l >= 1; --l)  // backward: the dX GEMM into layer l-1 closes one if layer l-1 has BatchNorm
    if (m->L[l - 1].bn) {
      if (cur == seg) { *floats = 2 * (size_t)m->L[l - 1].n_out; return LIPASR_OK; }
      ++cur;
    }
  return LIPASR_OK;
}

int lipasr_mlp_part_floats(lipasr_mlp_t m, size_t* floats) {
  LP_CHECK_ARG(m && floats, "lipasr_mlp_part_floats: null argument");
  *floats = 2 * (size_t)((m->max_batch + 31) / 32) * m->max_width;
  return LIPASR_OK;
}

int lipasr_mlp_train_segment(lipasr_mlp_t m, int seg, const float* params, float* bnstate, const float* x, const float* y_onehot,
                             int batch, float inv_batch, const lipasr_dropout_cfg* dropout, float* grads, float* loss_rows,
                             float* correct_rows, float* probs, float* part, int stat_batch, float stat_grad_scale,
                             lipasr_stream_t stream) {
  LP_CHECK_ARG(m != nullptr && part != nullptr, "lipasr_mlp_train_segment: null argument");
  int n = 0;
  (void)lipasr_mlp_train_segments(m, &n);
  LP_CHECK_ARG(seg >= 0 && seg < n, "lipasr_mlp_train_segment: segment %d outside [0, %d)", seg, n);
  LP_CHECK_ARG(stat_batch >= batch && stat_grad_scale > 0.0f, "lipasr_mlp_train_segment: stat_batch=%d (< batch %d) or scale %g", stat_batch,
               batch, (double)stat_grad_scale);
  SegArgs sa;
  sa.seg = seg; sa.part_ext = part; sa.stat_batch = stat_batch; sa.grad_scale = stat_grad_scale;
  return train_fwd_bwd_impl(m, params, bnstate, x, y_onehot, batch, inv_batch, dropout, grads, loss_rows, correct_rows, probs,
                            stream, 0, sa);
}

int lipasr_mlp_grad_split(lipasr_mlp_t m, size_t* late_floats) {
  LP_CHECK_ARG(m && late_floats, "lipasr_mlp_grad_split: null argument");
  // [W0 | b0] come from lipasr_mlp_train_dw0; what follows in the flat layout (gamma0, beta0, layers 1..) is final after the head
  const MlpLayer& L = m->L[0];
  *late_floats = (m->n_layers == 1) ? 0 : (L.bn ? L.offg : m->L[1].offW);
  return LIPASR_OK;
}

int lipasr_mlp_predict(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x, int batch,
                       float* probs, float* logits, lipasr_stream_t stream) {
  int rc = check_batch("lipasr_mlp_predict", m, batch);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(params && x && (probs || logits), "lipasr_mlp_predict: null argument");
  LP_CHECK_ARG(m->n_state == 0 || bnstate, "lipasr_mlp_predict: bnstate is null");
  float* lg = logits ? logits : (m->ws + m->offLogits);
  rc = forward_infer(m, params, bnstate, x, batch, false, lg, S(stream));
  if (rc != LIPASR_OK) return rc;
  if (probs) {
    const int C = m->L[m->n_layers - 1].n_out;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((batch + 255) / 256), dim3(256), 0, S(stream), lg, (const float*)nullptr,
                       batch, C, 0.0f, probs, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    LP_LAUNCH_CHECK();
  }
  return LIPASR_OK;
}

int lipasr_mlp_own_labels(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x, int batch,
                          float* y_onehot_out, lipasr_stream_t stream) {
  int rc = check_batch("lipasr_mlp_own_labels", m, batch);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(params && x && y_onehot_out, "lipasr_mlp_own_labels: null argument");
  LP_CHECK_ARG(m->n_state == 0 || bnstate, "lipasr_mlp_own_labels: bnstate is null");
  float* lg = m->ws + m->offLogits;
  rc = forward_infer(m, params, bnstate, x, batch, false, lg, S(stream));
  if (rc != LIPASR_OK) return rc;
  const int C = m->L[m->n_layers - 1].n_out;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3((batch + 255) / 256), dim3(256), 0, S(stream), lg, (const float*)nullptr,
                     batch, C, 0.0f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, y_onehot_out);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

static int attack_common(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x_eval,
                         const float* y_onehot, int batch, float* dx, float* x_adv, const float* x0, float alpha,
                         float eps, hipStream_t st) {
  float* lg = m->ws + m->offLogits;
  bool fused = false;
  int rc = forward_infer(m, params, bnstate, x_eval, batch, true, lg, st, y_onehot, &fused);
  if (rc != LIPASR_OK) return rc;
  if (!fused) {
    const int C = m->L[m->n_layers - 1].n_out;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, lg, y_onehot, batch, C,
                       1.0f / (float)batch, m->ws + m->offProb, m->ws + m->offDzLast, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr);
    LP_LAUNCH_CHECK();
  }
  return backward_infer(m, params, bnstate, batch, dx, x_adv, x0, alpha, eps, st, 1.0f / (float)batch);
}

int lipasr_mlp_input_grad(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x,
                          const float* y_onehot, int batch, float* dx, lipasr_stream_t stream) {
  int rc = check_batch("lipasr_mlp_input_grad", m, batch);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(params && x && y_onehot && dx, "lipasr_mlp_input_grad: null argument");
  LP_CHECK_ARG(m->n_state == 0 || bnstate, "lipasr_mlp_input_grad: bnstate is null");
  return attack_common(m, params, bnstate, x, y_onehot, batch, dx, nullptr, nullptr, 0.0f, 0.0f, S(stream));
}

int lipasr_mlp_attack_step(lipasr_mlp_t m, const float* params, const float* bnstate, float* x_adv, const float* x0,
                           const float* y_onehot, int batch, float alpha, float eps, lipasr_stream_t stream) {
  int rc = check_batch("lipasr_mlp_attack_step", m, batch);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(params && x_adv && x0 && y_onehot, "lipasr_mlp_attack_step: null argument");
  LP_CHECK_ARG(m->n_state == 0 || bnstate, "lipasr_mlp_attack_step: bnstate is null");
  LP_CHECK_ARG(eps >= 0.0f && !(alpha != alpha), "lipasr_mlp_attack_step: eps=%g alpha=%g", (double)eps, (double)alpha);
  return attack_common(m, params, bnstate, x_adv, y_onehot, batch, nullptr, x_adv, x0, alpha, eps, S(stream));
}

int lipasr_mlp_output_vjp(lipasr_mlp_t m, const float* params, const float* bnstate, const float* x, const float* v,
                          int on_logits, int batch, float* probs_out, float* dx, lipasr_stream_t stream) {
  int rc = check_batch("lipasr_mlp_output_vjp", m, batch);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(params && x && v && dx, "lipasr_mlp_output_vjp: null argument");
  LP_CHECK_ARG(m->n_state == 0 || bnstate, "lipasr_mlp_output_vjp: bnstate is null");
  hipStream_t st = S(stream);
  float* lg = m->ws + m->offLogits;
  rc = forward_infer(m, params, bnstate, x, batch, true, lg, st);
  if (rc != LIPASR_OK) return rc;
  const int C = m->L[m->n_layers - 1].n_out;
  hipLaunchKernelGGL(softmax_vjp_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, lg, v, batch, C, on_logits ? 1 : 0,
                     probs_out, m->ws + m->offDzLast);
  LP_LAUNCH_CHECK();
  return backward_infer(m, params, bnstate, batch, dx, nullptr, nullptr, 0.0f, 0.0f, st);
}

int lipasr_mlp_set_gemm_tiles(lipasr_mlp_t m, int lds_min_tiles) {
  LP_CHECK_ARG(m != nullptr && lds_min_tiles >= 0, "lipasr_mlp_set_gemm_tiles: bad argument");
  m->lds_min_tiles = lds_min_tiles;
  return LIPASR_OK;
}

int lipasr_mlp_set_fuse_bn(lipasr_mlp_t m, int mode) {
  LP_CHECK_ARG(m != nullptr && (mode == 0 || mode == 1), "lipasr_mlp_set_fuse_bn: bad argument");
  m->fuse_bn = mode;
  return LIPASR_OK;
}

int lipasr_mlp_set_cu_budget(lipasr_mlp_t m, int n_cus) {
  LP_CHECK_ARG(m != nullptr && n_cus >= 0, "lipasr_mlp_set_cu_budget: bad argument");
  m->cu_budget = n_cus;
  return LIPASR_OK;
}

int lipasr_mlp_exchange_errors(lipasr_mlp_t m, int* errors_host) {
  LP_CHECK_ARG(m != nullptr && errors_host != nullptr, "lipasr_mlp_exchange_errors: null argument");
  *errors_host = 0;
  if (!m->xc_err) return LIPASR_OK;
  DeviceGuard g(m->ctx->device);
  LP_HIP(hipMemcpy(errors_host, m->xc_err, sizeof(int), hipMemcpyDeviceToHost));  // synchronises with the device
  if (*errors_host) LP_HIP(hipMemset(m->xc_err, 0, sizeof(int)));
  return LIPASR_OK;
}

int lipasr_mlp_set_compute(lipasr_mlp_t m, int mode) {
  LP_CHECK_ARG(m != nullptr, "lipasr_mlp_set_compute: null plan");
  LP_CHECK_ARG(mode >= 0 && mode <= 2, "lipasr_mlp_set_compute: mode %d (0 = exact fp32, 1 = bf16 operands, 2 = fp16 two-plane split)", mode);
  if (mode == 2) {
    // the split needs operands inside fp16's range: gradients carry a measured scale, kernels are small, and the activations are
    // bounded because they are BatchNorm outputs -- so every hidden layer must have one (the reference's models do:
    // train_constraints.py:67-85); a network without runs its activations up without bound (all-positive kernels: 1e4 and more)
    for (int l = 0; l + 1 < m->n_layers; ++l)
      if (!m->L[l].bn) {
        set_error("lipasr_mlp_set_compute: mode 2 (fp16 two-plane split) needs BatchNormalization after every hidden Dense layer (layer %d has none); use mode 0", l);
        return LIPASR_EUNSUPPORTED;
      }
  }
  m->compute_bf16 = mode;
  return LIPASR_OK;
}

}  // extern "C"
