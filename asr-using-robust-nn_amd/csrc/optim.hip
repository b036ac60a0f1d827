// K5 (Keras-form Adam + NonNeg over the flat parameter buffer), K4 (stand-alone sign step),
// A2 (StandardScaler fit / apply) and the plan-level wrappers of the K3 projections.
#include "mlp.h"

namespace lipasr {

struct SegTable {
  int n;
  unsigned start[4 * LIPASR_MAX_LAYERS];  // ascending, in floats, multiples of 4
  unsigned char nonneg[4 * LIPASR_MAX_LAYERS];
};

// One launch over the flat buffers.  t = *step_dev + 1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (Keras
// optimizer_v2.Adam, epsilon outside the square root and not bias-corrected).
__global__ __launch_bounds__(256) void adam_nonneg_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, size_t n4,
                                                           SegTable segs, const int* __restrict__ step_dev, float lr,
                                                           float b1, float b2, float eps, float gscale) {
  __shared__ float lr_t_s;
  if (threadIdx.x == 0) {
    const double t = (double)(*step_dev + 1);
    lr_t_s = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const float lr_t = lr_t_s;
  const float ob1 = 1.0f - b1, ob2 = 1.0f - b2;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const unsigned e = (unsigned)(i << 2);
    int s = 0;
    while (s + 1 < segs.n && e >= segs.start[s + 1]) ++s;
    const bool nn = segs.nonneg[s] != 0;
    float4 wv = reinterpret_cast<float4*>(w)[i];
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define LP_ADAM1(c)                                  \
  {                                                  \
    const float gg = gv.c * gscale;                  \
    mv.c = mv.c * b1 + gg * ob1;                     \
    vv.c = vv.c * b2 + (gg * gg) * ob2;              \
    float x = wv.c - lr_t * mv.c / (sqrtf(vv.c) + eps); \
    wv.c = (nn && !(x >= 0.0f)) ? 0.0f : x;          \
  }
    LP_ADAM1(x) LP_ADAM1(y) LP_ADAM1(z) LP_ADAM1(w)
#undef LP_ADAM1
    reinterpret_cast<float4*>(w)[i] = wv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

__global__ void step_inc_kernel(int* step_dev) { *step_dev += 1; }

__global__ __launch_bounds__(256) void sign_step_kernel(float* __restrict__ x_adv, const float* __restrict__ x0,
                                                         const float* __restrict__ g, size_t n, float alpha, float eps) {
  const size_t stride = (size_t)gridDim.x * 256;
  const bool noclip = isinf(eps);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gv = g[i];
    const float sg = (gv > 0.0f) ? 1.0f : ((gv < 0.0f) ? -1.0f : 0.0f);
    const float xa = x_adv[i] + alpha * sg;
    const float b = x0[i];
    x_adv[i] = noclip ? xa : b + fminf(fmaxf(xa - b, -eps), eps);
  }
}

// per-feature mean and population std in fp64, 32 columns x 8 row lanes per workgroup, two passes
__global__ __launch_bounds__(256) void scaler_fit_kernel(const float* __restrict__ x, int n_rows, int n_feat,
                                                          double* __restrict__ mean_out, double* __restrict__ scale_out) {
  __shared__ double part[8][33];
  const int tid = threadIdx.x, cx = tid & 31, ry = tid >> 5;
  const int j = blockIdx.x * 32 + cx;
  double s = 0.0;
  if (j < n_feat)
    for (int b = ry; b < n_rows; b += 8) s += (double)x[(size_t)b * n_feat + j];
  part[ry][cx] = s;
  __syncthreads();
  double tot = 0.0;
  for (int k = 0; k < 8; ++k) tot += part[k][cx];
  const double mean = tot / (double)n_rows;
  __syncthreads();
  double q = 0.0;
  if (j < n_feat)
    for (int b = ry; b < n_rows; b += 8) {
      const double d = (double)x[(size_t)b * n_feat + j] - mean;
      q += d * d;
    }
  part[ry][cx] = q;
  __syncthreads();
  if (ry == 0 && j < n_feat) {
    double tq = 0.0;
    for (int k = 0; k < 8; ++k) tq += part[k][cx];
    const double var = tq / (double)n_rows;
    double sc = sqrt(var);
    // sklearn _handle_zeros_in_scale: (near-)constant features keep scale 1
    if (sc < 10.0 * 2.220446049250313e-16 || var <= (double)n_rows * 2.220446049250313e-16 * var +
                                                        ((double)n_rows * mean * 2.220446049250313e-16) *
                                                            ((double)n_rows * mean * 2.220446049250313e-16))
      sc = 1.0;
    mean_out[j] = mean;
    scale_out[j] = sc;
  }
}

__global__ __launch_bounds__(256) void scaler_apply_kernel(const float* __restrict__ x, size_t n, int n_feat,
                                                            const double* __restrict__ mean, const double* __restrict__ scale,
                                                            float* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int j = (int)(i % (size_t)n_feat);
    out[i] = (float)(((double)x[i] - mean[j]) / scale[j]);
  }
}

static int layer_arrays(lipasr_mlp_t m, float* params, float** Ws, int* rows, int* cols) {
  for (int l = 0; l < m->n_layers; ++l) {
    Ws[l] = params + m->L[l].offW;
    rows[l] = m->L[l].n_in;
    cols[l] = m->L[l].n_out;
  }
  return m->n_layers;
}

}  // namespace lipasr

using namespace lipasr;

extern "C" {

static int adam_launch(lipasr_mlp_t m, float* params, const float* grads, float* adam_m, float* adam_v, int* step_dev,
                       float lr, float beta1, float beta2, float eps, float grad_scale, bool bump_step,
                       lipasr_stream_t stream) {
  LP_CHECK_ARG(m && params && grads && adam_m && adam_v && step_dev, "lipasr_mlp_adam_nonneg: null argument");
  LP_CHECK_ARG(((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) |
                 reinterpret_cast<uintptr_t>(adam_m) | reinterpret_cast<uintptr_t>(adam_v)) & 15) == 0,
               "lipasr_mlp_adam_nonneg: flat buffers must be 16-byte aligned");
  LP_CHECK_ARG(lr > 0.0f && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f,
               "lipasr_mlp_adam_nonneg: bad hyper-parameters");
  SegTable segs;
  memset(&segs, 0, sizeof(segs));
  int n = 0;
  for (int l = 0; l < m->n_layers; ++l) {
    const MlpLayer& L = m->L[l];
    segs.start[n] = (unsigned)L.offW; segs.nonneg[n++] = L.nonneg ? 1 : 0;
    segs.start[n] = (unsigned)L.offb; segs.nonneg[n++] = 0;
    if (L.bn) {
      segs.start[n] = (unsigned)L.offg; segs.nonneg[n++] = 0;
      segs.start[n] = (unsigned)L.offbe; segs.nonneg[n++] = 0;
    }
  }
  segs.n = n;
  const size_t n4 = m->n_params >> 2;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_nonneg_kernel, dim3(blocks), dim3(256), 0, S(stream), params, grads, adam_m, adam_v, n4, segs,
                     step_dev, lr, beta1, beta2, eps, grad_scale);
  LP_LAUNCH_CHECK();
  if (bump_step) {
    // a second one-thread launch: a last-workgroup ticket inside the Adam kernel measured 11 us slower (atomics
    // from 1-2 K workgroups on one address) than this 4.6 us launch
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, S(stream), step_dev);
    LP_LAUNCH_CHECK();
  }
  return LIPASR_OK;
}

int lipasr_mlp_adam_nonneg(lipasr_mlp_t m, float* params, const float* grads, float* adam_m, float* adam_v,
                           int* step_dev, float lr, float beta1, float beta2, float eps, float grad_scale,
                           lipasr_stream_t stream) {
  return adam_launch(m, params, grads, adam_m, adam_v, step_dev, lr, beta1, beta2, eps, grad_scale, true, stream);
}

int lipasr_mlp_adam_project_product(lipasr_mlp_t m, float* params, const float* grads, float* adam_m, float* adam_v,
                                    int* step_dev, float lr, float beta1, float beta2, float eps, float grad_scale,
                                    float rho, const int* order, int n_order, float* norms_out,
                                    lipasr_stream_t stream) {
  int rc = adam_launch(m, params, grads, adam_m, adam_v, step_dev, lr, beta1, beta2, eps, grad_scale, false, stream);
  if (rc != LIPASR_OK) return rc;
  float* Ws[LIPASR_MAX_LAYERS];
  int rows[LIPASR_MAX_LAYERS], cols[LIPASR_MAX_LAYERS];
  const int n = layer_arrays(m, params, Ws, rows, cols);
  // the step counter moves inside the projection's single-workgroup kernel (after Adam has read it)
  return lipasr::project_product_bump(m->ctx, Ws, rows, cols, n, rho, order, n_order, norms_out, step_dev, stream);
}

int lipasr_sign_step(lipasr_handle_t h, float* x_adv, const float* x0, const float* g, size_t n, float alpha, float eps,
                     lipasr_stream_t stream) {
  LP_CHECK_ARG(h && x_adv && x0 && g, "lipasr_sign_step: null argument");
  LP_CHECK_ARG(eps >= 0.0f, "lipasr_sign_step: eps=%g must be non-negative", (double)eps);
  if (n == 0) return LIPASR_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sign_step_kernel, dim3(blocks), dim3(256), 0, S(stream), x_adv, x0, g, n, alpha, eps);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_scaler_fit(lipasr_handle_t h, const float* x, int n_rows, int n_feat, double* mean_out, double* scale_out,
                      lipasr_stream_t stream) {
  LP_CHECK_ARG(h && x && mean_out && scale_out, "lipasr_scaler_fit: null argument");
  LP_CHECK_ARG(n_rows >= 1 && n_feat >= 1, "lipasr_scaler_fit: empty data %dx%d", n_rows, n_feat);
  hipLaunchKernelGGL(scaler_fit_kernel, dim3((n_feat + 31) / 32), dim3(256), 0, S(stream), x, n_rows, n_feat, mean_out,
                     scale_out);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_scaler_apply(lipasr_handle_t h, const float* x, int n_rows, int n_feat, const double* mean,
                        const double* scale, float* out, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && x && mean && scale && out, "lipasr_scaler_apply: null argument");
  LP_CHECK_ARG(n_rows >= 0 && n_feat >= 1, "lipasr_scaler_apply: bad shape %dx%d", n_rows, n_feat);
  const size_t n = (size_t)n_rows * n_feat;
  if (n == 0) return LIPASR_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scaler_apply_kernel, dim3(blocks), dim3(256), 0, S(stream), x, n, n_feat, mean, scale, out);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_mlp_project_product(lipasr_mlp_t m, float* params, float rho, const int* order, int n_order,
                               float* norms_out, lipasr_stream_t stream) {
  LP_CHECK_ARG(m && params, "lipasr_mlp_project_product: null argument");
  float* Ws[LIPASR_MAX_LAYERS];
  int rows[LIPASR_MAX_LAYERS], cols[LIPASR_MAX_LAYERS];
  const int n = layer_arrays(m, params, Ws, rows, cols);
  return lipasr_project_product(m->ctx, Ws, rows, cols, n, rho, order, n_order, norms_out, stream);
}

int lipasr_mlp_project_per_layer(lipasr_mlp_t m, float* params, float rho, float* v_state, int warm, int iters,
                                 float* sigmas_out, lipasr_stream_t stream) {
  LP_CHECK_ARG(m && params, "lipasr_mlp_project_per_layer: null argument");
  float* Ws[LIPASR_MAX_LAYERS];
  int rows[LIPASR_MAX_LAYERS], cols[LIPASR_MAX_LAYERS];
  const int n = layer_arrays(m, params, Ws, rows, cols);
  return lipasr_project_per_layer(m->ctx, Ws, rows, cols, n, rho, v_state, warm, iters, sigmas_out, stream);
}

int lipasr_mlp_product_norm(lipasr_mlp_t m, const float* params, float* sigma_out, lipasr_stream_t stream) {
  LP_CHECK_ARG(m && params, "lipasr_mlp_product_norm: null argument");
  float* Ws[LIPASR_MAX_LAYERS];
  int rows[LIPASR_MAX_LAYERS], cols[LIPASR_MAX_LAYERS];
  const int n = layer_arrays(m, const_cast<float*>(params), Ws, rows, cols);
  return lipasr_product_norm(m->ctx, Ws, rows, cols, n, sigma_out, stream);
}

}  // extern "C"
