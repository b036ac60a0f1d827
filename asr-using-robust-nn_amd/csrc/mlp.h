// The dense-classifier plan shared by dense.hip (forward/backward) and optim.hip (Adam, projections).
#pragma once
#include "common.h"

namespace lipasr {

struct MlpLayer {
  int n_in = 0, n_out = 0;
  bool bn = false, nonneg = false;
  float dropout = 0.f;
  // offsets in floats into the flat trainable buffer (params / grads / adam m, v)
  size_t offW = 0, offb = 0, offg = 0, offbe = 0;
  // offsets in floats into bnstate
  size_t offmm = 0, offmv = 0;
  // offsets in floats into the plan workspace
  size_t offA = 0;     // post-ReLU activations [max_batch][n_out]            (hidden layers)
  size_t offH = 0;     // post-BN/dropout activations = next layer's input     (== offA if neither)
  size_t offMean = 0;  // saved batch mean [n_out], rstd follows at offMean + n_out (BN layers)
  size_t offDz = 0;    // gradient at this layer's pre-activation [max_batch][n_out] (kept for the grouped dW launch)
};

constexpr float kBnMomentum = 0.99f;  // Keras BatchNormalization defaults (train_constraints.py:68)
constexpr float kBnEps = 1e-3f;

}  // namespace lipasr

struct lipasr_mlp {
  lipasr_ctx* ctx = nullptr;
  int n_layers = 0;
  int max_batch = 0;
  int max_width = 0;
  lipasr::MlpLayer L[LIPASR_MAX_LAYERS];
  size_t n_params = 0, n_state = 0;
  float* ws = nullptr;  // workspace
  size_t ws_floats = 0;
  size_t offLogits = 0, offProb = 0, offDzLast = 0, offG0 = 0, offG1 = 0, offG2 = 0, offPart = 0;
  int compute_bf16 = 0;  // lipasr_mlp_set_compute: 0 exact fp32, 1 GEMM operands rounded to bf16 at the MFMA, 2 fp16 two-plane split (fp32 accumulate)
  float last_inv_batch = 1.0f;  // the loss-gradient bound of the last training forward (mode 2's gradient scale in lipasr_mlp_train_dw0)
  int lds_min_tiles = 0;  // lipasr_mlp_set_gemm_tiles: training GEMMs take the LDS-tiled kernel from this many 64x64 tiles (0 = default)
  // Round 5: training-mode BatchNorm inside the GEMM that produces its input (dense.hip, "exchange epilogue").  The row tiles
  // of a 32- (or 64-) column block hand each other their column partial sums through memory INSIDE the launch, so the apply
  // kernels and their launch boundaries go.  Per BatchNorm layer and direction: granules {tag, value} and two control words
  // per 32-column block.
  int fuse_bn = 1;        // lipasr_mlp_set_fuse_bn: 0 = the launch chain (GEMM + apply kernel), the parity reference
  int cu_budget = 0;      // lipasr_mlp_set_cu_budget: CUs the stream this plan runs on may use (0 = all of the device)
  int n_cus = 0;
  int xc_rt_max = 0;      // row tiles the granule regions hold (<= 64)
  unsigned long long* xc_gran = nullptr;
  unsigned* xc_ctrl = nullptr;
  unsigned* amax = nullptr;  // device words [layer]: max |dz_layer| of the current step as float bits (arithmetic mode 2's gradient scales)
  int* xc_err = nullptr;  // device word: an exchange gave up (a workgroup of its column block never became resident)
  size_t xc_gran_off[2][LIPASR_MAX_LAYERS] = {};  // [forward | backward][layer], in granules
  size_t xc_ctrl_off[2][LIPASR_MAX_LAYERS] = {};  // in 32-bit words
};
