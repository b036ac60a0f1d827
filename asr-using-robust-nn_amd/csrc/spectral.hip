// K3: Lipschitz projections on the device (gfx950).
//
//   product variant  (simple_norm_constraint, Constraints.py:135-189; get_lipschitz_constrained,
//                     extract_features_construct_dataset.py:169-196)
//     cst = W_m^T ... W_1^T is R x n_0 with R = number of classes (<= 32).  It is formed by m-1
//     skinny chain steps  P <- P * W_k^T  (each output element is a dot product along a contiguous
//     row of W_k: one wavefront per row, 64-lane DPP/shuffle reduction, W read exactly once), the
//     last step also emits per-workgroup partial Gram matrices P P^T in fp64.  One small kernel
//     sums the partials in a fixed order, takes lambda_max by repeated squaring
//     (lambda_max <= tr(G^(2^J))^(1/2^J) <= R^(1/2^J) lambda_max, J = 40), derives every scale factor
//     of the sequential pass in closed form and writes them to scratch; a last multi-tensor kernel
//     applies them.  No host round trip, bitwise reproducible (no float atomics), so data-parallel
//     replicas stay identical.
//
//   per-layer variant (norm_constraint, Constraints.py:9-33; get_norms,
//                      extract_features_construct_dataset.py:154-161)
//     power iteration on W^T W, all layers batched in each launch, warm-started from the previous
//     step's right singular vector.  u = W v: one wavefront per row; v = W^T u: 32 columns x 8 row
//     lanes per workgroup with an LDS tree, fixed summation order.
#include "common.h"

namespace lipasr {

constexpr double kEps = 2.220446049250313e-16;  // np.spacing(1), Constraints.py:25,167
constexpr int kMaxR = 32;
constexpr int kMaxOrder = 64;
constexpr int kChainRowsPerWave = 1;
constexpr int kChainRowsPerBlock = 4 * kChainRowsPerWave;
#ifndef LIPASR_SQUARINGS
#define LIPASR_SQUARINGS 40
#endif
constexpr int kSquarings = LIPASR_SQUARINGS;
#ifndef LIPASR_SIGMA_WGS
#define LIPASR_SIGMA_WGS 192  // workgroups of sigma_scale_layers_kernel (each recomputes sigma, then scales its slice of a layer)
#endif

// ---------------------------------------------------------------------------------------------
// chain step: P_out[r][i] = sum_j P_in[r][j] * W[i][j],  i < n_rows, j < n_in, r < R
//   p_mode 0: P_in is R x n_in row-major
//   p_mode 1: P_in[r][j] = Wlast[j*R + r]   (P = W_m^T read straight from the last kernel)
//   p_mode 2: P_in = identity (R == n_in)    (single-layer model: P_out = W^T)
// emit_gram: also write this block's partial Gram  sum_i P_out[:,i] P_out[:,i]^T  (fp64, R*R)
// ---------------------------------------------------------------------------------------------
template <int RM>  // compile-time bound on R: keeps the unrolled body (and the instruction footprint) small
__global__ __launch_bounds__(256) void chain_step_kernel(const float* __restrict__ Pin, int p_mode,
                                                          const float* __restrict__ W, int n_rows, int n_in, int R,
                                                          float* __restrict__ Pout, int emit_gram,
                                                          double* __restrict__ gram_part) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ps = reinterpret_cast<float*>(smem_raw);                      // [R][n_in]
  float* rowsum = Ps + (size_t)R * n_in;                               // [4][32]
  double* gsm = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(rowsum + 4 * kMaxR) + 7) & ~uintptr_t(7));  // [4][R*R]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = R * n_in;
  // this wavefront's weight row starts its trip from L2/HBM before P is staged (first 1024 columns in registers)
  const bool vec = ((n_in & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  const int i_pref = blockIdx.x * kChainRowsPerBlock + wave * kChainRowsPerWave;
  float4 wpre[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = lane * 4 + 256 * t;
    wpre[t] = (vec && i_pref < n_rows && j < n_in) ? *reinterpret_cast<const float4*>(W + (size_t)i_pref * n_in + j)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (p_mode == 0) {
    if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(Pin) & 15) == 0) {
      const float4* src = reinterpret_cast<const float4*>(Pin);
      float4* dst = reinterpret_cast<float4*>(Ps);
      const int t4 = total >> 2;
#pragma unroll 4
      for (int f = tid; f < t4; f += 256) dst[f] = src[f];
    } else {
      for (int f = tid; f < total; f += 256) Ps[f] = Pin[f];
    }
  } else if (p_mode == 1) {
    for (int f = tid; f < total; f += 256) {
      int j = f / R, r = f - j * R;
      Ps[r * n_in + j] = Pin[f];
    }
  } else {
    for (int f = tid; f < total; f += 256) {
      int r = f / n_in, j = f - r * n_in;
      Ps[f] = (r == j) ? 1.0f : 0.0f;
    }
  }
  const int RR = R * R;
  double gacc[(RM * RM + 63) / 64];
#pragma unroll
  for (int q = 0; q < (RM * RM + 63) / 64; ++q) gacc[q] = 0.0;
  __syncthreads();

  for (int rr = 0; rr < kChainRowsPerWave; ++rr) {
    const int i = blockIdx.x * kChainRowsPerBlock + wave * kChainRowsPerWave + rr;
    const bool live = i < n_rows;  // wave-uniform
    float acc[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) acc[r] = 0.0f;
    if (live) {
      const float* wrow = W + (size_t)i * n_in;
      if (vec) {
        int tt = 0;
        for (int j = lane * 4; j < n_in; j += 256, ++tt) {
          float4 w;
          if (rr == 0 && tt < 4) w = (tt == 0) ? wpre[0] : (tt == 1) ? wpre[1] : (tt == 2) ? wpre[2] : wpre[3];
          else w = *reinterpret_cast<const float4*>(wrow + j);
#pragma unroll
          for (int r = 0; r < RM; ++r) {
            if (r < R) {
              const float4 p = *reinterpret_cast<const float4*>(Ps + r * n_in + j);
              acc[r] = fmaf(w.x, p.x, acc[r]);
              acc[r] = fmaf(w.y, p.y, acc[r]);
              acc[r] = fmaf(w.z, p.z, acc[r]);
              acc[r] = fmaf(w.w, p.w, acc[r]);
            }
          }
        }
      } else {
        for (int j = lane; j < n_in; j += 64) {
          const float w = wrow[j];
#pragma unroll
          for (int r = 0; r < RM; ++r)
            if (r < R) acc[r] = fmaf(w, Ps[r * n_in + j], acc[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RM; ++r)
      if (r < R) acc[r] = wave_sum(acc[r]);
    if (live && lane == 0) {
#pragma unroll
      for (int r = 0; r < RM; ++r)
        if (r < R) Pout[(size_t)r * n_rows + i] = acc[r];
    }
    if (emit_gram) {
      // publish this row's R sums to the wave's LDS slot, then each lane adds its Gram entries
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RM; ++r)
          if (r < R) rowsum[wave * kMaxR + r] = live ? acc[r] : 0.0f;
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
#pragma unroll
      for (int q = 0; q < (RM * RM + 63) / 64; ++q) {
        const int e = lane + 64 * q;
        if (e < RR) {
          const int a = e / R, b = e - a * R;
          gacc[q] += (double)rowsum[wave * kMaxR + a] * (double)rowsum[wave * kMaxR + b];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (emit_gram) {
#pragma unroll
    for (int q = 0; q < (RM * RM + 63) / 64; ++q) {
      const int e = lane + 64 * q;
      if (e < RR) gsm[wave * RR + e] = gacc[q];
    }
    __syncthreads();
    for (int e = tid; e < RR; e += 256)
      gram_part[(size_t)blockIdx.x * RR + e] = (gsm[e] + gsm[RR + e]) + (gsm[2 * RR + e] + gsm[3 * RR + e]);
  }
}

// ---------------------------------------------------------------------------------------------
// The first (small) chain steps in ONE launch.  The chain starts at the narrow end of the network: W_m^T is R x 64, the
// next kernels are 128 x 64, 256 x 128, 512 x 256 -- 0.7 MB of weights and 1.7 M multiply-adds that cost three dependent
// launches of 6-7 us each (round trips, not work).  Here every workgroup (512 threads) recomputes the leading products
// P_1 ... P_(n-1) in full in its own LDS and then its own 128 rows of the last fused step, which it writes out; the remaining
// (large) steps run as chain_step_kernel launches on that output.  Recomputing is only cheap on the matrix pipe: a step is
// P_next^T [rows x R] = W [rows x n_in] . P^T [n_in x R], one v_mfma_f32_16x16x4_f32 tile per 16 weight rows (exact fp32, an fmaf
// chain along k; R <= 16 in the tile's columns), the weight fragments of a tile loaded in one go and the next tile's -- of the next
// step too, they do not depend on P -- fetched while this one multiplies.  (A first version kept chain_step_kernel's
// arithmetic -- a wavefront per row, ten wave reductions per row -- and was bitwise equal to the separate launches, but
// 400 rows x 170 instructions per workgroup took 90 us; round 3's version ran all rows through ONE workgroup: +192 us.)
// The sums are associated differently from chain_step_kernel's, so the product differs from the per-step launches in the
// last bits (1e-7); every replica runs the same code.
// ---------------------------------------------------------------------------------------------
constexpr int kHeadMax = 3;
constexpr int kHeadQ = 16;   // k-blocks of 16 per tile row: n_in <= 256
typedef float head_f32x4 __attribute__((ext_vector_type(4)));
struct ChainHead {
  int n;                      // fused steps (>= 2)
  const float* W[kHeadMax];   // step s: n_rows[s] x n_in[s], row-major, n_in a multiple of 16
  int n_rows[kHeadMax];
  int n_in[kHeadMax];         // n_in[s + 1] == n_rows[s]
  const float* Wlast;         // P_0[r][j] = Wlast[j * R + r]
  int R;                      // <= 16
  float* Pout;                // R x n_rows[n - 1]
  int rows_per_block;         // of the last fused step (128: one tile per wavefront)
};

__device__ __forceinline__ void head_load(const float* __restrict__ W, int n_rows, int n_in, int i0, int lane, float4 (&a)[kHeadQ]) {
  const int row = i0 + (lane & 15);
  const float* p = W + (size_t)min(row, n_rows - 1) * n_in + 4 * (lane >> 4);
#pragma unroll
  for (int q = 0; q < kHeadQ; ++q) {
    if (16 * q < n_in) {
      a[q] = *reinterpret_cast<const float4*>(p + 16 * q);
      if (row >= n_rows) a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// one 16-row tile: acc[reg] = P_next[r = lane & 15][i0 + 4 (lane >> 4) + reg].  T: [16][n_in + 4] in LDS, rows >= R zero
__device__ __forceinline__ head_f32x4 head_tile(const float4 (&a)[kHeadQ], const float* __restrict__ T, int n_in, int lane) {
  head_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* t = T + (lane & 15) * (n_in + 4) + 4 * (lane >> 4);
#pragma unroll
  for (int q = 0; q < kHeadQ; ++q) {
    if (16 * q < n_in) {
      const float4 b = *reinterpret_cast<const float4*>(t + 16 * q);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b.w, acc, 0, 0, 0);
    }
  }
  return acc;
}

__global__ __launch_bounds__(512) void chain_head_kernel(ChainHead a) {
  extern __shared__ __attribute__((aligned(16))) float hs[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = a.R;
  // the tile sequence of this wavefront: step s < n - 1: tiles wave, wave + 8, ... of n_rows[s] / 16; last step: the tiles of this
  // workgroup's rows.  The first tile's weights start their trip before P_0 is staged.
  float4 cur[kHeadQ], nxt[kHeadQ];
  const int last_i0 = blockIdx.x * a.rows_per_block, last_i1 = min(a.n_rows[a.n - 1], last_i0 + a.rows_per_block);
  auto first_tile = [&](int s) { return (s + 1 < a.n) ? 16 * wave : last_i0 + 16 * wave; };
  auto end_row = [&](int s) { return (s + 1 < a.n) ? a.n_rows[s] : last_i1; };
  if (first_tile(0) < end_row(0)) head_load(a.W[0], a.n_rows[0], a.n_in[0], first_tile(0), lane, cur);
  float* Pin = hs;
  {
    const int ld = a.n_in[0] + 4;
    for (int f = tid; f < 16 * ld; f += 512) Pin[f] = 0.0f;
    __syncthreads();
    const int total = R * a.n_in[0];
    for (int f = tid; f < total; f += 512) {
      const int j = f / R, r = f - j * R;
      Pin[r * ld + j] = a.Wlast[f];
    }
  }
  __syncthreads();
  for (int s = 0; s < a.n; ++s) {
    const int n_in = a.n_in[s], n_rows = a.n_rows[s];
    const bool last = (s + 1 == a.n);
    float* Pnext = Pin + 16 * (n_in + 4);
    const int ldn = n_rows + 4;
    if (!last) {  // rows >= R of the next panel stay zero (the tile's columns R .. 15 multiply them)
      for (int f = tid + R * ldn; f < 16 * ldn; f += 512) Pnext[f] = 0.0f;
    }
    const int e = end_row(s);
    for (int i0 = first_tile(s); i0 < e; i0 += 128) {
      // the weights of this wavefront's next tile -- of this step, or the first one of the next step
      const bool more = i0 + 128 < e;
      if (more) head_load(a.W[s], n_rows, n_in, i0 + 128, lane, nxt);
      else if (!last && first_tile(s + 1) < end_row(s + 1)) head_load(a.W[s + 1], a.n_rows[s + 1], a.n_in[s + 1], first_tile(s + 1), lane, nxt);
      const head_f32x4 acc = head_tile(cur, Pin, n_in, lane);
      const int r = lane & 15, ib = i0 + 4 * (lane >> 4);
      if (r < R) {
        if (!last) {
          *reinterpret_cast<float4*>(Pnext + r * ldn + ib) = make_float4(acc[0], acc[1], acc[2], acc[3]);  // (rows past n_rows: zero weights -> zeros, inside the padding)
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (ib + q < n_rows) a.Pout[(size_t)r * n_rows + ib + q] = acc[q];
        }
      }
#pragma unroll
      for (int q = 0; q < kHeadQ; ++q) cur[q] = nxt[q];
    }
    if (!last) {
      if (!(first_tile(s) < e) && first_tile(s + 1) < end_row(s + 1))  // this wavefront had no tile in step s: nobody fetched its next one
        head_load(a.W[s + 1], a.n_rows[s + 1], a.n_in[s + 1], first_tile(s + 1), lane, cur);
      __syncthreads();
      Pin = Pnext;
    }
  }
}

// x^(1/m) for x > 0 and a small integer m (the number of layers), to fp64 accuracy without the fp64 pow() (a serial chain of
// thousands of cycles, six of them in a row per projection): range reduction by frexp, a hardware fp32 log2 / exp2 estimate of
// the root of the mantissa part (1e-7), two Newton steps on y^m = f (1e-14, 1e-28).
__device__ __forceinline__ double root_m(double x, int m) {
  if (!(x > 0.0) || !(x < 1.7e308)) return pow(x, 1.0 / (double)m);  // (zero, negative, inf, nan: the library's answers)
  int e;
  double f = frexp(x, &e);  // x = f 2^e, f in [0.5, 1)
  int q = e / m, r = e - q * m;
  if (r < 0) { r += m; q -= 1; }
  f = ldexp(f, r);  // f in [0.5, 2^(m-1)], x = f 2^(q m)
  double y = (double)exp2f(log2f((float)f) / (float)m);
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    double ym = y;
    for (int k = 1; k < m; ++k) ym *= y;      // y^m
    y -= y * (1.0 - f / ym) / (double)m;
  }
  return ldexp(y, q);
}

struct OrderArgs {
  int n_layers;
  int n_order;
  int order[kMaxOrder];
};

// One workgroup: G = sum of partial Grams; sigma = sqrt(lambda_max(G)); closed-form sequential scales.
constexpr int kSigmaThreads = 1024;
constexpr int kSliceDoubles = 1024;  // LDS for the sliced partial sums: min(32, 1024 / R^2) slices x R^2 entries

__global__ __launch_bounds__(kSigmaThreads) void product_sigma_kernel(const double* __restrict__ gram_part, int n_part,
                                                                       int R, double rho, OrderArgs oa,
                                                                       float* __restrict__ scales,
                                                                       float* __restrict__ norms_out,
                                                                       float* __restrict__ sigma_out,
                                                                       int* __restrict__ bump) {
  __shared__ double A[kMaxR * kMaxR];
  __shared__ double B[kMaxR * kMaxR];
  __shared__ double slice_sum[kSliceDoubles];
  __shared__ double tr_s;
  const int tid = threadIdx.x;
  const int RR = R * R;
  // Gram = sum of the per-workgroup partials.  The partials were written by other XCDs, so every load is a trip to
  // memory: thread (slice, entry) takes partials slice, slice + n_sl, ... with eight loads in flight, fixed order.
  {
    int n_sl = kSigmaThreads / RR;  // R <= 32, so at least one
    if (n_sl > 32) n_sl = 32;
    if (n_sl > n_part) n_sl = n_part;
    const int sl = tid / RR, ent = tid - sl * RR;
    if (sl < n_sl) {
      double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      for (int p = sl; p < n_part; p += 8 * n_sl) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int q = p + u * n_sl;
          acc[u] += (q < n_part) ? gram_part[(size_t)q * RR + ent] : 0.0;
        }
      }
      slice_sum[sl * RR + ent] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    __syncthreads();
    for (int e = tid; e < RR; e += kSigmaThreads) {
      double s = 0.0;
      for (int k = 0; k < n_sl; ++k) s += slice_sum[k * RR + e];
      A[e] = s;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int a = 0; a < R; ++a) t += A[a * R + a];
    tr_s = t;
  }
  __syncthreads();
  const double t0 = tr_s;
  double log_lambda = 0.0;
  bool zero = !(t0 > 0.0);
  if (!zero) {
    log_lambda = log(t0);
    for (int e = tid; e < RR; e += kSigmaThreads) A[e] /= t0;
    __syncthreads();
    double wgt = 0.5;
    // A <- A^2 / tr(A^2), ONE barrier per squaring and no in-place write: the buffers keep the UNnormalised square and the
    // factor 1 / tr of the previous round is applied (squared) when the next square is formed.  Every thread forms the trace
    // itself from a buffer nobody writes until all threads have passed the next barrier (round 4's form rescaled `nxt` in
    // place right after the trace reads, with no barrier between them: a wavefront that was late reading the diagonal saw a
    // mix of scaled and unscaled entries -- ADVICE r4).
    double* cur = A;
    double* nxt = B;
    double c2 = 1.0;  // (1 / trace of `cur`)^2; A was normalised above
    for (int it = 0; it < kSquarings; ++it) {
      for (int e = tid; e < RR; e += kSigmaThreads) {
        const int a = e / R, b = e - a * R;
        double s = 0.0;
        for (int c = 0; c < R; ++c) s = fma(cur[a * R + c], cur[c * R + b], s);
        nxt[e] = s * c2;
      }
      __syncthreads();
      double tj = 0.0;
      for (int a = 0; a < R; ++a) tj += nxt[a * R + a];
      log_lambda += wgt * log(tj);
      wgt *= 0.5;
      const double inv = 1.0 / tj;
      c2 = inv * inv;
      double* t = cur; cur = nxt; nxt = t;
      if (fabs(tj - 1.0) < 1e-12) break;  // rank-1 projector reached (same value in every thread): the squarings left would add < 1e-12 2^-it to log lambda; at 1e-15 rounding noise in the trace kept the loop running all 40 rounds
    }
  }
  if (tid == 0) {
    if (bump) *bump += 1;  // the optimizer's step counter, when this projection closes a fused Adam + projection step
    const double sigma = zero ? 0.0 : exp(0.5 * log_lambda);
    if (sigma_out) *sigma_out = (float)sigma;
    if (scales) {
      double sc[LIPASR_MAX_LAYERS];
      for (int l = 0; l < oa.n_layers; ++l) sc[l] = 1.0;
      double n = sigma;
      for (int v = 0; v < oa.n_order; ++v) {
        norms_out[v] = (float)n;
        const double s = root_m(rho / (n + kEps), oa.n_layers);
        sc[oa.order[v]] *= s;
        n *= s;
      }
      norms_out[oa.n_order] = (float)n;
      for (int l = 0; l < oa.n_layers; ++l) scales[l] = (float)sc[l];
    }
  }
}

struct LayerPtrs {
  int n_layers;
  float* W[LIPASR_MAX_LAYERS];
  int rows[LIPASR_MAX_LAYERS];
  int cols[LIPASR_MAX_LAYERS];
  int wg_start[LIPASR_MAX_LAYERS + 1];  // sigma_scale_layers_kernel only: workgroups wg_start[l] .. wg_start[l + 1] - 1 scale layer l
};

// blockIdx.y = layer; W_l *= scales[l] (skipped when the factor is exactly 1)
__global__ __launch_bounds__(256) void scale_layers_kernel(LayerPtrs lp, const float* __restrict__ scales) {
  const int l = blockIdx.y;
  const float s = scales[l];
  if (s == 1.0f) return;
  float* w = lp.W[l];
  const size_t n = (size_t)lp.rows[l] * lp.cols[l];
  const size_t stride = (size_t)gridDim.x * 256;
  if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
    const size_t n4 = n >> 2;
    float4* w4 = reinterpret_cast<float4*>(w);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      float4 x = w4[i];
      x.x *= s; x.y *= s; x.z *= s; x.w *= s;
      w4[i] = x;
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) w[i] *= s;
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) w[i] *= s;
  }
}

// Round 4: the eigenvalue step and the rescaling in ONE launch.  Every workgroup recomputes sigma for itself from the finished
// product P (R x n, 35 kB for the reference's model: staged in LDS, Gram in fp64, the same normalised squarings as
// product_sigma_kernel) and then scales its slice of its layer -- identical arithmetic on identical inputs in every workgroup,
// so all of them apply the same factors.  Replaces product_sigma_kernel (14 us: one workgroup summing 220 cross-XCD Gram
// partials) + a launch boundary + scale_layers_kernel (8 us) on the critical path of every training step; the chain's last
// step no longer emits Gram partials for it.  Workgroup (0, 0) also writes scales / norms / sigma and bumps the step counter.
#ifndef LIPASR_SSL_THREADS
#define LIPASR_SSL_THREADS 512
#endif
constexpr int kSslThreads = LIPASR_SSL_THREADS, kSslWaves = kSslThreads / 64;  // 256 / 512 / 1024 threads: 22.8 / 20.6 / 29.6 us (the Gram of the product, 4.8 us with four wavefronts, gets faster, everything behind a barrier slower)
__global__ __launch_bounds__(kSslThreads) void sigma_scale_layers_kernel(const float* __restrict__ P, int n, int R, double rho, OrderArgs oa,
                                                                  LayerPtrs lp, float* __restrict__ scales_out, float* __restrict__ norms_out,
                                                                  float* __restrict__ sigma_out, int* __restrict__ bump) {
  extern __shared__ __attribute__((aligned(16))) char ssl_raw[];
  const int tid = threadIdx.x;
  const int RR = R * R, total = R * n;
  // dynamic LDS: A | B | four Gram partials (doubles, R^2 each) | P [R][n] floats: 40 kB for 10 x 880, four workgroups per CU
  double* A = reinterpret_cast<double*>(ssl_raw);
  double* B = A + RR;
  double* gpart0 = B + RR;  // [kSslWaves][RR]
  float* Ps = reinterpret_cast<float*>(gpart0 + kSslWaves * RR);
  __shared__ float sc_s[LIPASR_MAX_LAYERS];
  if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(P) & 15) == 0) {
    const float4* src = reinterpret_cast<const float4*>(P);
    float4* dst = reinterpret_cast<float4*>(Ps);
    for (int f = tid; f < (total >> 2); f += kSslThreads) dst[f] = src[f];
  } else {
    for (int f = tid; f < total; f += kSslThreads) Ps[f] = P[f];
  }
  __syncthreads();
  // Gram (symmetric: the R (R + 1) / 2 entries a <= b): wavefront w takes the column quads q = w (mod 4), lane e (and e + 64, ...)
  // the entry; float4 reads of both rows, fp64 products and sums in ascending column order
  {
    const int w = tid >> 6, lane = tid & 63;
    const int n_ent = R * (R + 1) / 2, nq = n >> 2;
    for (int e = lane; e < n_ent; e += 64) {
      int a = 0, rem = e;
      while (rem >= R - a) { rem -= R - a; ++a; }  // e -> (a, b = a + rem)
      const int b = a + rem;
      const float* pa = Ps + (size_t)a * n;
      const float* pb = Ps + (size_t)b * n;
      double s = 0.0;
      if ((n & 3) == 0) {
        for (int q = w; q < nq; q += kSslWaves) {
          const float4 x = *reinterpret_cast<const float4*>(pa + 4 * q), y = *reinterpret_cast<const float4*>(pb + 4 * q);
          s = fma((double)x.x, (double)y.x, s); s = fma((double)x.y, (double)y.y, s);
          s = fma((double)x.z, (double)y.z, s); s = fma((double)x.w, (double)y.w, s);
        }
      } else {
        for (int i = w; i < n; i += kSslWaves) s = fma((double)pa[i], (double)pb[i], s);
      }
      gpart0[w * RR + a * R + b] = s;
      gpart0[w * RR + b * R + a] = s;
    }
  }
  __syncthreads();
  for (int e = tid; e < RR; e += kSslThreads) {  // (fixed order: a pairwise tree over the wavefronts' partial sums)
    double part[kSslWaves];
#pragma unroll
    for (int w = 0; w < kSslWaves; ++w) part[w] = gpart0[w * RR + e];
#pragma unroll
    for (int h = kSslWaves / 2; h >= 1; h >>= 1)
#pragma unroll
      for (int w = 0; w < h; ++w) part[w] = part[2 * w] + part[2 * w + 1];
    A[e] = part[0];
  }
  __syncthreads();
  double t0 = 0.0;
  for (int a = 0; a < R; ++a) t0 += A[a * R + a];
  __syncthreads();
  __shared__ double tj_s[kSquarings + 1];
  __shared__ double ll_s;
  const bool zero = !(t0 > 0.0);
  int n_it = 0;
  if (!zero) {
    for (int e = tid; e < RR; e += kSslThreads) A[e] /= t0;
    __syncthreads();
    // A <- A^2 / tr(A^2).  The traces are kept and their logarithms taken AFTER the loop, all at once (one fp64 log per
    // squaring inside the loop was a serial chain of ~150 dependent instructions in every workgroup: 6 of the kernel's 22 us)
    // One barrier per squaring and no in-place write (see product_sigma_kernel): the buffers keep the unnormalised square, the
    // previous round's 1 / trace is applied, squared, where the next square is formed; a buffer is rewritten only after every
    // thread has passed the barrier behind its last reader.
    double* cur = A;
    double* nxt = B;
    double c2 = 1.0;
    for (int it = 0; it < kSquarings; ++it) {
      for (int e = tid; e < RR; e += kSslThreads) {
        const int a = e / R, b = e - a * R;
        double s = 0.0;
        for (int c = 0; c < R; ++c) s = fma(cur[a * R + c], cur[c * R + b], s);
        nxt[e] = s * c2;
      }
      __syncthreads();
      double tj = 0.0;
      for (int a = 0; a < R; ++a) tj += nxt[a * R + a];
      if (tid == 0) tj_s[it] = tj;
      n_it = it + 1;
      const double inv = 1.0 / tj;
      c2 = inv * inv;
      double* t = cur; cur = nxt; nxt = t;
      if (fabs(tj - 1.0) < 1e-12) break;  // (the same value in every thread of every workgroup; 1e-15 sat inside the trace's rounding noise: all 40 rounds ran)
    }
    // log lambda = log t0 + sum_j 2^-(j+1) log t_j: thread j < n_it takes term j, wavefront 0 sums them in a fixed order
    if (tid < 64) {
      double term = 0.0;
      if (tid < n_it) term = ldexp(log(tj_s[tid]), -(tid + 1));
      term = wave_sum_d(term);
      if (tid == 0) ll_s = log(t0) + term;
    }
    __syncthreads();
  }
  const double log_lambda = zero ? 0.0 : ll_s;
  if (tid == 0) {
    const bool first = blockIdx.x == 0;
    if (first && bump) *bump += 1;
    const double sigma = zero ? 0.0 : exp(0.5 * log_lambda);
    if (first && sigma_out) *sigma_out = (float)sigma;
    double sc[LIPASR_MAX_LAYERS];
    for (int l = 0; l < oa.n_layers; ++l) sc[l] = 1.0;
    double nrm = sigma;
    for (int v = 0; v < oa.n_order; ++v) {
      if (first) norms_out[v] = (float)nrm;
      const double s = root_m(rho / (nrm + kEps), oa.n_layers);
      sc[oa.order[v]] *= s;
      nrm *= s;
    }
    if (first) norms_out[oa.n_order] = (float)nrm;
    for (int l = 0; l < oa.n_layers; ++l) {
      sc_s[l] = (float)sc[l];
      if (first && scales_out) scales_out[l] = (float)sc[l];
    }
  }
  __syncthreads();
  int l = 0;
  while (l + 1 < lp.n_layers && (int)blockIdx.x >= lp.wg_start[l + 1]) ++l;
  const float s = sc_s[l];
  if (s == 1.0f) return;
  float* w = lp.W[l];
  const size_t nw = (size_t)lp.rows[l] * lp.cols[l];
  const size_t j = (size_t)((int)blockIdx.x - lp.wg_start[l]);
  const size_t stride = (size_t)(lp.wg_start[l + 1] - lp.wg_start[l]) * kSslThreads;
  if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
    const size_t n4 = nw >> 2;
    float4* w4 = reinterpret_cast<float4*>(w);
    for (size_t i = j * kSslThreads + threadIdx.x; i < n4; i += stride) {
      float4 x = w4[i];
      x.x *= s; x.y *= s; x.z *= s; x.w *= s;
      w4[i] = x;
    }
    for (size_t i = (n4 << 2) + j * kSslThreads + threadIdx.x; i < nw; i += stride) w[i] *= s;
  } else {
    for (size_t i = j * kSslThreads + threadIdx.x; i < nw; i += stride) w[i] *= s;
  }
}

// ---------------------------------------------------------------------------------------------
// power iteration, all layers per launch
// ---------------------------------------------------------------------------------------------
struct PiArgs {
  int n_layers;
  const float* W[LIPASR_MAX_LAYERS];
  int rows[LIPASR_MAX_LAYERS];
  int cols[LIPASR_MAX_LAYERS];
  float* v[LIPASR_MAX_LAYERS];   // [cols]
  float* u[LIPASR_MAX_LAYERS];   // [rows]
  int blk_u[LIPASR_MAX_LAYERS + 1];  // block prefix for the u kernel (8 rows per block)
  int blk_v[LIPASR_MAX_LAYERS + 1];  // block prefix for the v kernel (32 columns per block)
  int clamp;
};

constexpr int kPiMaxDim = 8192;

__device__ __forceinline__ int find_layer(const int* prefix, int n, int blk) {
  int l = 0;
  while (l + 1 < n && blk >= prefix[l + 1]) ++l;
  return l;
}

__device__ __forceinline__ float start_vec(int j) {
  // fixed, strictly positive start vector (Perron direction for the clamped kernels)
  return 1.0f + 0.001f * (float)((((unsigned)j * 2654435761u) >> 24) & 255u);
}

__device__ __forceinline__ float block_sum_256(float x, float* red) {
  x = wave_sum(x);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = x;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// u = max(W,0)? * (v / ||v||)
__global__ __launch_bounds__(256) void pi_u_kernel(PiArgs a, int cold) {
  __shared__ __attribute__((aligned(16))) float vs[kPiMaxDim];
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = find_layer(a.blk_u, a.n_layers, blockIdx.x);
  const int rows = a.rows[l], cols = a.cols[l];
  const float* W = a.W[l];
  float ss = 0.0f;
  for (int j = tid; j < cols; j += 256) {
    const float x = cold ? start_vec(j) : a.v[l][j];
    vs[j] = x;
    ss = fmaf(x, x, ss);
  }
  const float nv2 = block_sum_256(ss, red);
  const float inv = nv2 > 0.0f ? 1.0f / sqrtf(nv2) : 0.0f;
  const bool vec = ((cols & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  const int row0 = (blockIdx.x - a.blk_u[l]) * 8 + wave * 2;
  for (int rr = 0; rr < 2; ++rr) {
    const int i = row0 + rr;
    if (i >= rows) break;  // wave-uniform
    const float* wrow = W + (size_t)i * cols;
    float acc = 0.0f;
    if (vec) {
      for (int j = lane * 4; j < cols; j += 256) {
        float4 w = *reinterpret_cast<const float4*>(wrow + j);
        const float4 p = *reinterpret_cast<const float4*>(vs + j);
        if (a.clamp) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
        acc = fmaf(w.x, p.x, acc); acc = fmaf(w.y, p.y, acc); acc = fmaf(w.z, p.z, acc); acc = fmaf(w.w, p.w, acc);
      }
    } else {
      for (int j = lane; j < cols; j += 64) {
        float w = wrow[j];
        if (a.clamp) w = fmaxf(w, 0.f);
        acc = fmaf(w, vs[j], acc);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) a.u[l][i] = acc * inv;
  }
}

// v = max(W,0)?^T u   (unnormalised; the next pi_u normalises)
__global__ __launch_bounds__(256) void pi_v_kernel(PiArgs a) {
  __shared__ float us[kPiMaxDim];
  __shared__ float part[8][33];
  const int tid = threadIdx.x, cx = tid & 31, ry = tid >> 5;
  const int l = find_layer(a.blk_v, a.n_layers, blockIdx.x);
  const int rows = a.rows[l], cols = a.cols[l];
  const float* W = a.W[l];
  for (int i = tid; i < rows; i += 256) us[i] = a.u[l][i];
  __syncthreads();
  const int j = (blockIdx.x - a.blk_v[l]) * 32 + cx;
  float acc = 0.0f;
  if (j < cols) {
#pragma unroll 4
    for (int i = ry; i < rows; i += 8) {
      float w = W[(size_t)i * cols + j];
      if (a.clamp) w = fmaxf(w, 0.f);
      acc = fmaf(w, us[i], acc);
    }
  }
  part[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && j < cols) {
    float s = ((part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx])) +
              ((part[4][cx] + part[5][cx]) + (part[6][cx] + part[7][cx]));
    a.v[l][j] = s;
  }
}

// sigma_l = ||u_l|| (u = W v_hat); optionally W_l <- max(W_l,0) * c / (sigma_l + eps).  blockIdx.y = layer.
__global__ __launch_bounds__(256) void pi_finish_kernel(PiArgs a, double c, int do_scale, float* __restrict__ sigmas_out) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int l = blockIdx.y;
  const int rows = a.rows[l], cols = a.cols[l];
  float ss = 0.0f;
  for (int i = tid; i < rows; i += 256) {
    const float x = a.u[l][i];
    ss = fmaf(x, x, ss);
  }
  const float s2 = block_sum_256(ss, red);
  const float sigma = sqrtf(s2);
  if (blockIdx.x == 0 && tid == 0) sigmas_out[l] = sigma;
  if (!do_scale) return;
  const double f = c / ((double)sigma + kEps);
  float* w = const_cast<float*>(a.W[l]);
  const size_t n = (size_t)rows * cols;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += stride) {
    const float x = fmaxf(w[i], 0.0f);
    w[i] = (float)((double)x * f);
  }
}

// ---------------------------------------------------------------------------------------------
// Frobenius projection (customConstraint) and BN correction factor
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_clamped_kernel(const float* __restrict__ w, size_t n, double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0.0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float x = fmaxf(w[i], 0.0f);
    s += (double)x * (double)x;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void frob_scale_kernel(float* __restrict__ w, size_t n, const double* __restrict__ part,
                                                          int n_part, double rho) {
  __shared__ double tot;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int p = 0; p < n_part; ++p) s += part[p];
    tot = s;
  }
  __syncthreads();
  const float f = (float)(rho / (sqrt(tot) + kEps));
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) w[i] = fmaxf(w[i], 0.0f) * f;
}

__global__ __launch_bounds__(256) void bn_correction_kernel(const float* __restrict__ gamma, const float* __restrict__ var,
                                                             int n, float* __restrict__ out) {
  __shared__ float red[4];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, sqrtf(var[i]) / gamma[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int check_layers(const char* fn, const float* const* Ws, const int* rows, const int* cols, int n_layers,
                        bool need_chain) {
  LP_CHECK_ARG(Ws && rows && cols, "%s: null layer arrays", fn);
  LP_CHECK_ARG(n_layers >= 1 && n_layers <= LIPASR_MAX_LAYERS, "%s: n_layers=%d outside [1,%d]", fn, n_layers,
               LIPASR_MAX_LAYERS);
  for (int l = 0; l < n_layers; ++l) {
    LP_CHECK_ARG(Ws[l] != nullptr, "%s: kernel %d is null", fn, l);
    LP_CHECK_ARG(rows[l] >= 1 && cols[l] >= 1, "%s: kernel %d has shape %dx%d", fn, l, rows[l], cols[l]);
    if (need_chain && l > 0)
      LP_CHECK_ARG(rows[l] == cols[l - 1], "%s: kernel %d has %d inputs but kernel %d has %d outputs", fn, l, rows[l],
                   l - 1, cols[l - 1]);
  }
  return LIPASR_OK;
}

// Scratch layout (floats): [0, 64)       scales / sigma
//                          [64, ...)      two ping-pong P buffers, then the fp64 Gram partials
struct ChainScratch {
  float* scales;
  float* sigma;
  float* P[2];
  double* gram;
  size_t gram_cap_doubles;
};

static int carve_scratch(lipasr_ctx* h, int R, int max_width, ChainScratch* cs) {
  const size_t pbuf = ((size_t)R * max_width + 63) & ~size_t(63);
  const size_t need = 64 + 2 * pbuf;
  if (need + 1024 > h->scratch_floats) {
    set_error("product chain: %d classes x width %d exceeds the handle's scratch", R, max_width);
    return LIPASR_EUNSUPPORTED;
  }
  cs->scales = h->scratch;
  cs->sigma = h->scratch + 32;
  cs->P[0] = h->scratch + 64;
  cs->P[1] = h->scratch + 64 + pbuf;
  cs->gram = reinterpret_cast<double*>(h->scratch + 64 + 2 * pbuf);
  cs->gram_cap_doubles = (h->scratch_floats - (64 + 2 * pbuf)) / 2;
  return LIPASR_OK;
}

static int g_chain_head = -1;  // lipasr_debug_chain_head: -1 the two leading steps when legal, 0 none, n > 0 at most n (<= 3)

// Launches the chain; on return the Gram partials are in cs.gram (n_part blocks).
// want_gram = false: the caller takes the finished product itself (*p_final, R x rows[0]) and the last step emits nothing else.
static int launch_chain(lipasr_ctx* h, const float* const* Ws, const int* rows, const int* cols, int m,
                        ChainScratch* cs, int* n_part_out, hipStream_t st, bool want_gram = true, const float** p_final = nullptr) {
  const int R = cols[m - 1];
  if (R > kMaxR) {
    set_error("product chain: last layer has %d outputs; the Gram eigen-solver handles at most %d", R, kMaxR);
    return LIPASR_EUNSUPPORTED;
  }
  int max_width = 0;
  for (int l = 0; l < m; ++l) max_width = rows[l] > max_width ? rows[l] : max_width;
  int rc = carve_scratch(h, R, max_width, cs);
  if (rc != LIPASR_OK) return rc;

  const float* pin = Ws[m - 1];
  int p_mode = 1;
  int cur = 0;
  if (m == 1) {
    // P = W_1^T: run one step against the identity so the Gram falls out of the same kernel
    pin = nullptr;
    p_mode = 2;
  }
  int first_k = (m == 1) ? 0 : m - 2;
  // the leading small steps as one launch (chain_head_kernel): steps first_k ... first_k - n + 1, never the last step (k = 0,
  // which may emit the Gram partials); R <= 16, panels of <= 256 columns in multiples of 16
  if (m >= 4 && g_chain_head != 0 && R <= 16) {
    int n = 0;
    while (n < kHeadMax && first_k - n >= 1 && cols[first_k - n] <= 16 * kHeadQ && (cols[first_k - n] & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(Ws[first_k - n]) & 15) == 0)
      ++n;
    // measured (MI355X, reference widths, rocprofv3): two fused steps 7.2 us against 2 x 6.2, three 15.5 against 3 x 6.3 (the third
    // step's 96 extra matrix instructions per wavefront cost 8 us on the 4 CUs the launch occupies): two unless told otherwise
    if (n > (g_chain_head > 0 ? g_chain_head : 2)) n = (g_chain_head > 0 ? g_chain_head : 2);
    if (n >= 2) {
      ChainHead a;
      memset(&a, 0, sizeof(a));
      a.n = n;
      size_t lds_f = 0;
      for (int s2 = 0; s2 < n; ++s2) {
        a.W[s2] = Ws[first_k - s2];
        a.n_rows[s2] = rows[first_k - s2];
        a.n_in[s2] = cols[first_k - s2];
        lds_f += (size_t)16 * (a.n_in[s2] + 4);
      }
      a.Wlast = Ws[m - 1];
      a.R = R;
      a.Pout = cs->P[cur];
      a.rows_per_block = 128;
      const int blocks = (a.n_rows[n - 1] + a.rows_per_block - 1) / a.rows_per_block;
      const size_t lds = lds_f * sizeof(float);
      if (lds > 48 * 1024)
        LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(chain_head_kernel, dim3(blocks), dim3(512), lds, st, a);
      LP_LAUNCH_CHECK();
      pin = cs->P[cur];
      p_mode = 0;
      cur ^= 1;
      first_k -= n;
    }
  }
  for (int k = first_k; k >= 0; --k) {
    // step k multiplies by W_k^T: rows of W_k are the new columns of P
    const int n_rows = rows[k];
    const int n_in = (m == 1) ? R : cols[k];
    const int emit = (k == 0 && want_gram) ? 1 : 0;
    const int blocks = (n_rows + kChainRowsPerBlock - 1) / kChainRowsPerBlock;
    const size_t lds = (size_t)R * n_in * sizeof(float) + 4 * kMaxR * sizeof(float) +
                       (emit ? (size_t)4 * R * R * sizeof(double) : 0) + 16;
    if (lds > 160 * 1024 - 512) {
      set_error("product chain: %d x %d panel does not fit LDS", R, n_in);
      return LIPASR_EUNSUPPORTED;
    }
    if (emit && (size_t)blocks * R * R > cs->gram_cap_doubles) {
      set_error("product chain: Gram partials exceed scratch");
      return LIPASR_EUNSUPPORTED;
    }
    const float* Wk = (m == 1) ? Ws[0] : Ws[k];
#define LP_CHAIN(RM)                                                                                          \
  {                                                                                                           \
    if (lds > 48 * 1024)                                                                                      \
      LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_step_kernel<RM>),                        \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
    hipLaunchKernelGGL(chain_step_kernel<RM>, dim3(blocks), dim3(256), lds, st, pin, p_mode, Wk, n_rows, n_in, R, \
                       cs->P[cur], emit, cs->gram);                                                           \
  }
    if (R <= 12) LP_CHAIN(12) else if (R <= 20) LP_CHAIN(20) else LP_CHAIN(32)
#undef LP_CHAIN
    LP_LAUNCH_CHECK();
    if (emit) *n_part_out = blocks;
    if (k == 0 && p_final) *p_final = cs->P[cur];
    pin = cs->P[cur];
    p_mode = 0;
    cur ^= 1;
  }
  (void)h;
  return LIPASR_OK;
}

static int fill_pi_args(PiArgs* a, const float* const* Ws, const int* rows, const int* cols, int n_layers,
                        float* v_state, float* u_scratch, size_t u_cap, int clamp) {
  a->n_layers = n_layers;
  a->clamp = clamp;
  size_t voff = 0, uoff = 0;
  a->blk_u[0] = 0;
  a->blk_v[0] = 0;
  for (int l = 0; l < n_layers; ++l) {
    if (rows[l] > kPiMaxDim || cols[l] > kPiMaxDim) {
      set_error("power iteration: kernel %d is %dx%d; dimensions above %d are not supported", l, rows[l], cols[l],
                kPiMaxDim);
      return LIPASR_EUNSUPPORTED;
    }
    a->W[l] = Ws[l];
    a->rows[l] = rows[l];
    a->cols[l] = cols[l];
    a->v[l] = v_state + voff;
    a->u[l] = u_scratch + uoff;
    voff += cols[l];
    uoff += (rows[l] + 3) & ~3;
    a->blk_u[l + 1] = a->blk_u[l] + (rows[l] + 7) / 8;
    a->blk_v[l + 1] = a->blk_v[l] + (cols[l] + 31) / 32;
  }
  if (uoff > u_cap) {
    set_error("power iteration: left vectors exceed scratch");
    return LIPASR_EUNSUPPORTED;
  }
  return LIPASR_OK;
}

static int run_power_iteration(lipasr_ctx* h, PiArgs& a, int warm, int iters, hipStream_t st) {
  (void)h;
  const int nbu = a.blk_u[a.n_layers], nbv = a.blk_v[a.n_layers];
  int cold = warm ? 0 : 1;
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(pi_u_kernel, dim3(nbu), dim3(256), 0, st, a, cold);
    LP_LAUNCH_CHECK();
    cold = 0;
    hipLaunchKernelGGL(pi_v_kernel, dim3(nbv), dim3(256), 0, st, a);
    LP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(pi_u_kernel, dim3(nbu), dim3(256), 0, st, a, cold);
  LP_LAUNCH_CHECK();
  if (cold) {
    // iters == 0 on a cold start: still leave a defined warm-start vector behind
    hipLaunchKernelGGL(pi_v_kernel, dim3(nbv), dim3(256), 0, st, a);
    LP_LAUNCH_CHECK();
  }
  return LIPASR_OK;
}


// ---------------------------------------------------------------------------------------------
// Singular-value clipping of an R x n matrix, R <= 32 (norm_constraint_FISTA, Constraints.py:78-91).
// X = U S V^T  =>  U min(S, hi) V^T = M X  with  M = U diag(min(s, hi) / s) U^T, and U, S^2 are the
// eigen-pairs of the R x R Gram matrix X X^T.  One workgroup: Gram in fp64 through an LDS tile,
// cyclic Jacobi with the round-robin (tournament) ordering so that R/2 disjoint rotations run at
// once, then M X column by column.  Fixed summation order; no atomics.
// ---------------------------------------------------------------------------------------------
constexpr int kSvTile = 256;
constexpr int kSvLd = kMaxR + 1;
constexpr int kSvMaxSweeps = 30;

__global__ __launch_bounds__(256) void sv_clip_kernel(const float* __restrict__ X, int R, int n, double hi,
                                                       float* out, float* __restrict__ svals_out) {
  __shared__ double G[kMaxR * kSvLd];
  __shared__ double V[kMaxR * kSvLd];
  __shared__ double cs[kMaxR];  // (c, s) per pair
  __shared__ int pq[kMaxR];     // (p, q) per pair, p < q; -1 when the pair holds the bye
  __shared__ double red[kMaxR + 1];
  __shared__ float tile[kMaxR * (kSvTile + 1)];
  const int tid = threadIdx.x;
  const int RR = R * R;
  // ---- Gram
  double acc[(kMaxR * kMaxR) / 256] = {0.0, 0.0, 0.0, 0.0};
  for (int j0 = 0; j0 < n; j0 += kSvTile) {
    __syncthreads();
    for (int r = 0; r < R; ++r) tile[r * (kSvTile + 1) + tid] = (j0 + tid < n) ? X[(size_t)r * n + j0 + tid] : 0.0f;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (kMaxR * kMaxR) / 256; ++u) {
      const int e = tid + 256 * u;
      if (e < RR) {
        const int a = e / R, b = e - a * R;
        const float* ta = tile + a * (kSvTile + 1);
        const float* tb = tile + b * (kSvTile + 1);
        double s = 0.0;
        for (int c = 0; c < kSvTile; ++c) s = fma((double)ta[c], (double)tb[c], s);
        acc[u] += s;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < (kMaxR * kMaxR) / 256; ++u) {
    const int e = tid + 256 * u;
    if (e < RR) {
      const int a = e / R, b = e - a * R;
      G[a * kSvLd + b] = acc[u];
      V[a * kSvLd + b] = (a == b) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  // ---- Jacobi: G <- J^T G J, V <- V J
  const int Re = (R + 1) & ~1;  // players in the tournament (one bye when R is odd)
  const int n_pairs = Re / 2;
  for (int sweep = 0; sweep < kSvMaxSweeps; ++sweep) {
    // convergence: sum of squared off-diagonal entries against the squared diagonal
    if (tid < R) {
      double o = 0.0;
      for (int j = 0; j < R; ++j)
        if (j != tid) o = fma(G[tid * kSvLd + j], G[tid * kSvLd + j], o);
      red[tid] = o;
    }
    __syncthreads();
    double off = 0.0, dg = 0.0;
    for (int i = 0; i < R; ++i) {
      off += red[i];
      dg = fma(G[i * kSvLd + i], G[i * kSvLd + i], dg);
    }
    __syncthreads();
    if (!(off > 1e-30 * dg)) break;  // identical in every thread
    for (int round = 0; round < Re - 1; ++round) {
      if (tid < n_pairs) {
        int p, q;
        if (tid == 0) {
          p = Re - 1;
          q = round;
        } else {
          p = (round + tid) % (Re - 1);
          q = (round - tid + (Re - 1)) % (Re - 1);
        }
        if (p > q) { const int t = p; p = q; q = t; }
        double c = 1.0, sn = 0.0;
        if (q >= R) {
          p = -1;
        } else {
          const double gpp = G[p * kSvLd + p], gqq = G[q * kSvLd + q], gpq = G[p * kSvLd + q];
          if (fabs(gpq) > 1e-300 && fabs(gpq) > 1e-18 * sqrt(fabs(gpp * gqq))) {
            const double tau = (gqq - gpp) / (2.0 * gpq);
            const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            sn = t * c;
          }
        }
        pq[2 * tid] = p;
        pq[2 * tid + 1] = q;
        cs[2 * tid] = c;
        cs[2 * tid + 1] = sn;
      }
      __syncthreads();
      // columns p, q of G and of V
      for (int e = tid; e < n_pairs * R; e += 256) {
        const int k = e / R, i = e - k * R;
        const int p = pq[2 * k], q = pq[2 * k + 1];
        if (p < 0) continue;
        const double c = cs[2 * k], sn = cs[2 * k + 1];
        const double gip = G[i * kSvLd + p], giq = G[i * kSvLd + q];
        G[i * kSvLd + p] = c * gip - sn * giq;
        G[i * kSvLd + q] = sn * gip + c * giq;
        const double vip = V[i * kSvLd + p], viq = V[i * kSvLd + q];
        V[i * kSvLd + p] = c * vip - sn * viq;
        V[i * kSvLd + q] = sn * vip + c * viq;
      }
      __syncthreads();
      // rows p, q of G
      for (int e = tid; e < n_pairs * R; e += 256) {
        const int k = e / R, j = e - k * R;
        const int p = pq[2 * k], q = pq[2 * k + 1];
        if (p < 0) continue;
        const double c = cs[2 * k], sn = cs[2 * k + 1];
        const double gpj = G[p * kSvLd + j], gqj = G[q * kSvLd + j];
        G[p * kSvLd + j] = c * gpj - sn * gqj;
        G[q * kSvLd + j] = sn * gpj + c * gqj;
      }
      __syncthreads();
    }
  }
  // ---- singular values (descending) and the clipping factors
  if (tid < R) {
    const double lam = G[tid * kSvLd + tid];
    const double sv = (lam > 0.0) ? sqrt(lam) : 0.0;
    red[tid] = sv;
    cs[tid] = (sv > hi) ? hi / sv : 1.0;
  }
  __syncthreads();
  if (svals_out && tid < R) {
    const double mine = red[tid];
    int rank = 0;
    for (int k = 0; k < R; ++k) rank += (red[k] > mine || (red[k] == mine && k < tid)) ? 1 : 0;
    svals_out[rank] = (float)mine;
  }
  if (!out) return;
  // M = V diag(f) V^T, kept in G
  double mloc[(kMaxR * kMaxR) / 256];
#pragma unroll
  for (int u = 0; u < (kMaxR * kMaxR) / 256; ++u) {
    const int e = tid + 256 * u;
    mloc[u] = 0.0;
    if (e < RR) {
      const int a = e / R, b = e - a * R;
      double s = 0.0;
      for (int k = 0; k < R; ++k) s = fma(cs[k] * V[a * kSvLd + k], V[b * kSvLd + k], s);
      mloc[u] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < (kMaxR * kMaxR) / 256; ++u) {
    const int e = tid + 256 * u;
    if (e < RR) G[(e / R) * kSvLd + (e % R)] = mloc[u];
  }
  __syncthreads();
  // ---- out = M X, one column per thread (in place allowed: a tile is read before it is written)
  for (int j0 = 0; j0 < n; j0 += kSvTile) {
    const int j = j0 + tid;
    if (j < n) {
      float xcol[kMaxR];
#pragma unroll
      for (int b = 0; b < kMaxR; ++b) xcol[b] = (b < R) ? X[(size_t)b * n + j] : 0.0f;
      for (int a = 0; a < R; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < kMaxR; ++b)
          if (b < R) s = fma(G[a * kSvLd + b], (double)xcol[b], s);
        out[(size_t)a * n + j] = (float)s;
      }
    }
  }
}

}  // namespace lipasr

using namespace lipasr;

extern "C" {

int lipasr_debug_chain_head(int n) {
  g_chain_head = n < 0 ? -1 : (n > kHeadMax ? kHeadMax : n);
  return LIPASR_OK;
}

int lipasr_sigma_max(lipasr_handle_t h, const float* W, int rows, int cols, float* v_state, int warm, int iters,
                     int clamp_nonneg, float* sigma_out, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && W && v_state && sigma_out, "lipasr_sigma_max: null argument");
  LP_CHECK_ARG(rows >= 1 && cols >= 1 && iters >= 0, "lipasr_sigma_max: bad shape %dx%d or iters %d", rows, cols, iters);
  const float* Ws[1] = {W};
  PiArgs a;
  int rc = fill_pi_args(&a, Ws, &rows, &cols, 1, v_state, h->scratch + 64, h->scratch_floats - 64, clamp_nonneg);
  if (rc != LIPASR_OK) return rc;
  rc = run_power_iteration(h, a, warm, iters, S(stream));
  if (rc != LIPASR_OK) return rc;
  hipLaunchKernelGGL(pi_finish_kernel, dim3(1, 1), dim3(256), 0, S(stream), a, 0.0, 0, sigma_out);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_project_per_layer(lipasr_handle_t h, float* const* Ws, const int* rows, const int* cols, int n_layers,
                             float rho, float* v_state, int warm, int iters, float* sigmas_out,
                             lipasr_stream_t stream) {
  LP_CHECK_ARG(h && v_state && sigmas_out, "lipasr_project_per_layer: null argument");
  int rc = check_layers("lipasr_project_per_layer", Ws, rows, cols, n_layers, false);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(iters >= 0, "lipasr_project_per_layer: iters=%d", iters);
  LP_CHECK_ARG(rho > 0.0f, "lipasr_project_per_layer: rho=%g must be positive", (double)rho);
  PiArgs a;
  rc = fill_pi_args(&a, Ws, rows, cols, n_layers, v_state, h->scratch + 64, h->scratch_floats - 64, 1);
  if (rc != LIPASR_OK) return rc;
  rc = run_power_iteration(h, a, warm, iters, S(stream));
  if (rc != LIPASR_OK) return rc;
  const double c = pow((double)rho, 1.0 / (double)n_layers);  // np.power(rho, 1/self.m), Constraints.py:25
  hipLaunchKernelGGL(pi_finish_kernel, dim3(64, n_layers), dim3(256), 0, S(stream), a, c, 1, sigmas_out);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_project_product(lipasr_handle_t h, float* const* Ws, const int* rows, const int* cols, int n_layers,
                           float rho, const int* order, int n_order, float* norms_out, lipasr_stream_t stream) {
  return lipasr::project_product_bump(h, Ws, rows, cols, n_layers, rho, order, n_order, norms_out, nullptr, stream);
}

}  // extern "C"

// bump: optional device int incremented by the (single-workgroup) eigenvalue kernel -- lets the fused
// Adam + projection entry point drop its one-thread step-counter launch
int lipasr::project_product_bump(lipasr_handle_t h, float* const* Ws, const int* rows, const int* cols, int n_layers,
                                 float rho, const int* order, int n_order, float* norms_out, int* bump,
                                 lipasr_stream_t stream) {
  LP_CHECK_ARG(h && norms_out, "lipasr_project_product: null argument");
  int rc = check_layers("lipasr_project_product", Ws, rows, cols, n_layers, true);
  if (rc != LIPASR_OK) return rc;
  LP_CHECK_ARG(n_order >= 0 && n_order <= kMaxOrder, "lipasr_project_product: n_order=%d outside [0,%d]", n_order,
               kMaxOrder);
  LP_CHECK_ARG(n_order == 0 || order != nullptr, "lipasr_project_product: order is null");
  LP_CHECK_ARG(rho > 0.0f, "lipasr_project_product: rho=%g must be positive", (double)rho);
  OrderArgs oa;
  oa.n_layers = n_layers;
  oa.n_order = n_order;
  for (int v = 0; v < n_order; ++v) {
    LP_CHECK_ARG(order[v] >= 0 && order[v] < n_layers, "lipasr_project_product: order[%d]=%d out of range", v, order[v]);
    oa.order[v] = order[v];
  }
  ChainScratch cs;
  int n_part = 0;
  const int R = cols[n_layers - 1], n0 = rows[0];
  const size_t lds = (size_t)(2 + kSslWaves) * R * R * sizeof(double) + (size_t)R * n0 * sizeof(float) + 16;
  if (n_order > 0 && lds <= 96 * 1024) {
    // the step's critical path: chain -> ONE launch that finds sigma and rescales (every workgroup for itself)
    const float* p_final = nullptr;
    rc = launch_chain(h, Ws, rows, cols, n_layers, &cs, &n_part, S(stream), false, &p_final);
    if (rc != LIPASR_OK) return rc;
    LayerPtrs lp;
    lp.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) { lp.W[l] = Ws[l]; lp.rows[l] = rows[l]; lp.cols[l] = cols[l]; }
    static bool attr_set_dev[16] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_set = attr_set_dev[attr_dev & 15];
    if (lds > 32 * 1024 && !attr_set) {
      LP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sigma_scale_layers_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      attr_set = true;
    }
    // Workgroups in proportion to the layers' sizes (every one recomputes sigma first -- fp64, the same arithmetic on the same inputs --
    // and they share the CUs' fp64 rate: 64 per layer = 384 took 23.9 us on a 128-CU share, 128 per layer 38.7, 32 per layer 26.6 because
    // layer 1's slice got long)
    {
      size_t total = 0;
      for (int l = 0; l < n_layers; ++l) total += (size_t)rows[l] * cols[l];
      int acc_wg = 0;
      for (int l = 0; l < n_layers; ++l) {
        lp.wg_start[l] = acc_wg;
        const size_t sz = (size_t)rows[l] * cols[l];
        acc_wg += std::max(1, (int)((sz * LIPASR_SIGMA_WGS + total / 2) / total));
      }
      lp.wg_start[n_layers] = acc_wg;
    }
    hipLaunchKernelGGL(sigma_scale_layers_kernel, dim3(lp.wg_start[n_layers]), dim3(kSslThreads), lds, S(stream), p_final, n0, R, (double)rho, oa, lp, cs.scales,
                       norms_out, cs.sigma, bump);
    LP_LAUNCH_CHECK();
    return LIPASR_OK;
  }
  rc = launch_chain(h, Ws, rows, cols, n_layers, &cs, &n_part, S(stream));
  if (rc != LIPASR_OK) return rc;
  hipLaunchKernelGGL(product_sigma_kernel, dim3(1), dim3(kSigmaThreads), 0, S(stream), cs.gram, n_part, cols[n_layers - 1],
                     (double)rho, oa, cs.scales, norms_out, cs.sigma, bump);
  LP_LAUNCH_CHECK();
  if (n_order > 0) {
    LayerPtrs lp;
    lp.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) { lp.W[l] = Ws[l]; lp.rows[l] = rows[l]; lp.cols[l] = cols[l]; }
    hipLaunchKernelGGL(scale_layers_kernel, dim3(64, n_layers), dim3(256), 0, S(stream), lp, cs.scales);
    LP_LAUNCH_CHECK();
  }
  return LIPASR_OK;
}

extern "C" {

int lipasr_product_norm(lipasr_handle_t h, const float* const* Ws, const int* rows, const int* cols, int n_layers,
                        float* sigma_out, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && sigma_out, "lipasr_product_norm: null argument");
  int rc = check_layers("lipasr_product_norm", Ws, rows, cols, n_layers, true);
  if (rc != LIPASR_OK) return rc;
  ChainScratch cs;
  int n_part = 0;
  rc = launch_chain(h, Ws, rows, cols, n_layers, &cs, &n_part, S(stream));
  if (rc != LIPASR_OK) return rc;
  OrderArgs oa;
  oa.n_layers = n_layers;
  oa.n_order = 0;
  hipLaunchKernelGGL(product_sigma_kernel, dim3(1), dim3(kSigmaThreads), 0, S(stream), cs.gram, n_part, cols[n_layers - 1], 1.0,
                     oa, (float*)nullptr, (float*)nullptr, sigma_out, (int*)nullptr);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_frobenius_project(lipasr_handle_t h, float* W, size_t n, float rho, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && W && n > 0, "lipasr_frobenius_project: null or empty kernel");
  double* part = reinterpret_cast<double*>(h->scratch + 64);
  const int nb = 256;
  hipLaunchKernelGGL(sumsq_clamped_kernel, dim3(nb), dim3(256), 0, S(stream), W, n, part);
  LP_LAUNCH_CHECK();
  hipLaunchKernelGGL(frob_scale_kernel, dim3(nb), dim3(256), 0, S(stream), W, n, part, nb, (double)rho);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_bn_correction(lipasr_handle_t h, const float* gamma, const float* var, int n, float* out,
                         lipasr_stream_t stream) {
  LP_CHECK_ARG(h && gamma && var && out && n > 0, "lipasr_bn_correction: bad argument");
  hipLaunchKernelGGL(bn_correction_kernel, dim3(1), dim3(256), 0, S(stream), gamma, var, n, out);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_sv_clip(lipasr_handle_t h, const float* X, int R, int n, float hi, float* out, float* svals_out,
                   lipasr_stream_t stream) {
  LP_CHECK_ARG(h && X, "lipasr_sv_clip: null argument");
  LP_CHECK_ARG(out || svals_out, "lipasr_sv_clip: nothing to compute (out and svals_out both null)");
  LP_CHECK_ARG(R >= 1 && R <= kMaxR && n >= 1, "lipasr_sv_clip: needs 1 <= R <= 32 rows and n >= 1 columns");
  LP_CHECK_ARG(hi >= 0.0f, "lipasr_sv_clip: negative clipping level");
  hipLaunchKernelGGL(sv_clip_kernel, dim3(1), dim3(256), 0, S(stream), X, R, n, (double)hi, out, svals_out);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

}  // extern "C"
