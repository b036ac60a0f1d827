// Round 5, stage A of the persistent classifier step: what does ONE link of the narrow chain cost
//   GEMM(256 -> 128, bias, ReLU, column partial sums) -> BatchNorm apply -> GEMM(128 -> 64, ...)
// (a) as today's three launches (the library's own kernels, included below), against
// (b) inside one persistent launch whose workgroups form teams by the XCD they really run on (HW_REG_XCC_ID): a team owns
//     1/8 of the batch rows for every phase, so what one CU writes the next phase's CU reads from the SAME L2 (loads that
//     bypass L1: buffer_load ... sc1), behind a team barrier (one counter per team) instead of a kernel boundary, and
// (c) the price of the exchange a fused BatchNorm needs: every workgroup of a column block publishes its partial sums as
//     8-byte {tag, value} granules (sc1 stores), sweeps the granules of the block's other row tiles until every tag matches,
//     and the last one through bumps the block's epoch -- measured inside an otherwise empty kernel, R rounds per launch.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I asr-using-robust-nn_amd/csrc scratch/link_bench.hip -o scratch/link_bench
// Every spin is bounded (a give-up sets an error word, the grid always drains).
#include "../asr-using-robust-nn_amd/csrc/dense.hip"
#include <cstdlib>

namespace lipasr {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
}  // namespace lipasr

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);     \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int kB = 1024, kK1 = 256, kN1 = 128, kN2 = 64;
constexpr unsigned kSpinMax = 4000000u;  // x ~0.2 us: about a second, then give up

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct TeamState {
  unsigned count[8];      // members of team x (tickets), filled by the census
  unsigned census;        // workgroups that have registered
  unsigned err;           // a spin gave up
  unsigned pad[6];
  unsigned bar[8 * 32];   // team barrier counters, one 128-byte line each
};

// one 32 x 32 output tile = A[32 rows][K] (row-major, handed off inside the launch: loads bypass L1) x W[K][N] (read-only,
// plain loads), 4 wavefronts split K in 16-deep chunks as the library's gemm_tile does; bias + ReLU; per-tile column sums.
__device__ __forceinline__ void team_tile(const float* A, int lda, const float* __restrict__ W, int N, const float* __restrict__ bias,
                                          float* C, int K, int m0, int n0, float* part, float* red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0x7fffffff, 0x00020000);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  const int nch = K >> 4;
  for (int c = wave; c < nch; c += 4) {
    const int kb = c * 16 + 8 * h;
    const int off = ((m0 + r) * lda + kb) * 4;
    const i4v a_lo = __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 16);
    const i4v a_hi = __builtin_amdgcn_raw_buffer_load_b128(ra, off + 16, 0, 16);
    float b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = W[(size_t)(kb + q) * N + n0 + r];
    const float a[8] = {__int_as_float(a_lo.x), __int_as_float(a_lo.y), __int_as_float(a_lo.z), __int_as_float(a_lo.w),
                        __int_as_float(a_hi.x), __int_as_float(a_hi.y), __int_as_float(a_hi.z), __int_as_float(a_hi.w)};
#pragma unroll
    for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) red[wave * 1024 + ((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + r] = acc[q];
  __syncthreads();
  const int tcol = tid & 7, row = tid >> 3, c4 = tcol * 4;
  float4 s = *reinterpret_cast<const float4*>(red + row * 32 + c4);
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float4 t = *reinterpret_cast<const float4*>(red + w * 1024 + row * 32 + c4);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  const float4 bb = *reinterpret_cast<const float4*>(bias + n0 + c4);
  float4 o = make_float4(fmaxf(s.x + bb.x, 0.f), fmaxf(s.y + bb.y, 0.f), fmaxf(s.z + bb.z, 0.f), fmaxf(s.w + bb.w, 0.f));
  *reinterpret_cast<float4*>(C + (size_t)(m0 + row) * N + n0 + c4) = o;
  float cs[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int sft = 8; sft < 64; sft <<= 1) cs[e] += __shfl_xor(cs[e], sft, 64);
  __syncthreads();
  if (lane < 8)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave * 32 + lane * 4 + e] = cs[e];
  __syncthreads();
  if (tid < 32) part[(size_t)(m0 >> 5) * N + n0 + tid] = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
  __syncthreads();
}

// every storing wavefront drains, the workgroup meets, one lane arrives and polls; the other wavefronts leave through the barrier
__device__ __forceinline__ void team_barrier(TeamState* ts, int team, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* ctr = ts->bar + team * 32;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (ld_sc1(ctr) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > kSpinMax) { __hip_atomic_store(&ts->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}

struct LinkArgs {
  const float* x;   // [B][256]
  const float* W1;  // [256][128]
  const float* b1;
  const float* W2;  // [128][64]
  const float* b2;
  const float* gamma;
  const float* beta;
  float* a1;  // [B][128] post-ReLU
  float* h1;  // [B][128] after the apply
  float* a2;  // [B][64]
  float* part;  // [32][128] column partial sums (phase 1), reused per link
  TeamState* ts;
  int links;
  int n_wg;
};

__global__ __launch_bounds__(256) void team_links_kernel(LinkArgs p) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // 4 x 32 x 32 floats (+ padding that keeps one workgroup per CU)
  __shared__ unsigned s_team, s_idx, s_n;
  TeamState* ts = p.ts;
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    s_team = xcc;
    s_idx = __hip_atomic_fetch_add(&ts->count[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&ts->census, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (ld_sc1(&ts->census) < (unsigned)p.n_wg) {  // the census: every workgroup is resident and knows its team
      __builtin_amdgcn_s_sleep(4);
      if (++spins > kSpinMax) { __hip_atomic_store(&ts->err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    s_n = ld_sc1(&ts->count[xcc]);
  }
  __syncthreads();
  const int team = s_team, idx = s_idx, n = s_n;
  // slices: team x takes rows 128 x .. 128 x + 127 (a team without members would leave its slice undone: the host checks the
  // census and reports it; the product version deals the slices over the teams that exist)
  const int r0 = team * 128;
  unsigned epoch = 0;
  const __amdgpu_buffer_rsrc_t r_a1 = __builtin_amdgcn_make_buffer_rsrc(p.a1, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_part = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, 0x7fffffff, 0x00020000);
  for (int link = 0; link < p.links; ++link) {
    // phase 1: 4 x 4 tiles of the team's 128 x 128 slice of a1 = relu(x W1 + b1); link > 0 reads h1 columns as its input instead
    // (same shape of work: 256 input columns are x's; the dependency on the previous link is the team barrier)
    for (int t = idx; t < 16; t += n) team_tile(p.x, kK1, p.W1, kN1, p.b1, p.a1, kK1, r0 + 32 * (t >> 2), 32 * (t & 3), p.part, red);
    team_barrier(ts, team, (unsigned)n * (++epoch));
    // phase 2: BatchNorm apply on the slice with the statistics of the team's own four row tiles (timing stand-in for the
    // cross-team sums, which are priced separately): 128 rows x 128 columns = 4096 float4, spread over the team
    for (int f = idx * 256 + threadIdx.x; f < 4096; f += n * 256) {
      const int row = f >> 5, c4 = (f & 31) * 4;
      const i4v v = __builtin_amdgcn_raw_buffer_load_b128(r_a1, ((r0 + row) * kN1 + c4) * 4, 0, 16);
      float m[4], o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const i4v q = __builtin_amdgcn_raw_buffer_load_b128(r_part, (((r0 >> 5) + t) * kN1 + c4) * 4, 0, 16);
        m[0] += __int_as_float(q.x); m[1] += __int_as_float(q.y); m[2] += __int_as_float(q.z); m[3] += __int_as_float(q.w);
      }
      const float xv[4] = {__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w)};
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (xv[e] - m[e] * (1.0f / 128.0f)) * p.gamma[c4 + e] + p.beta[c4 + e];
      *reinterpret_cast<float4*>(p.h1 + (size_t)(r0 + row) * kN1 + c4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    team_barrier(ts, team, (unsigned)n * (++epoch));
    // phase 3: 4 x 2 tiles of a2 = relu(h1 W2 + b2)
    for (int t = idx; t < 8; t += n) team_tile(p.h1, kN1, p.W2, kN2, p.b2, p.a2, kN1, r0 + 32 * (t >> 1), 32 * (t & 1), p.part + 32 * 128, red);
    team_barrier(ts, team, (unsigned)n * (++epoch));
  }
}

// ---- (c) the granule exchange of a fused BatchNorm, alone
struct XchgArgs {
  u64* gran;        // [col blocks][row tiles][64] {tag, value}
  unsigned* epoch;  // [col blocks][32]: word 0 = epoch, word 1 = done counter
  unsigned* err;
  float* out;       // [col blocks][64]
  int row_tiles, rounds;
};

__global__ __launch_bounds__(256) void xchg_kernel(XchgArgs p) {
  __shared__ float vals[32 * 64];
  __shared__ unsigned s_e;
  const int cb = blockIdx.x, rt = blockIdx.y, tid = threadIdx.x;
  unsigned* ew = p.epoch + cb * 32;
  u64* g = p.gran + (size_t)cb * p.row_tiles * 64;
  for (int round = 0; round < p.rounds; ++round) {
    if (tid == 0) s_e = ld_sc1(ew);
    __syncthreads();
    const unsigned tag = s_e + 1;
    if (tid < 64) {
      const float v = (float)(rt + 1) * 0.5f + tid;
      __hip_atomic_store(g + (size_t)rt * 64 + tid, ((u64)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int total = p.row_tiles * 64;
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
      for (int i = tid; i < total; i += 256) {
        const u64 x = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (unsigned)(x >> 32) == tag;
        vals[i] = __uint_as_float((unsigned)x);
      }
      if (__syncthreads_and(ok)) break;
      if (++spins > kSpinMax / 64) { if (tid == 0) __hip_atomic_store(p.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    if (tid < 64) {
      double s = 0.0;
      for (int t = 0; t < p.row_tiles; ++t) s += (double)vals[t * 64 + tid];
      if (rt == 0) p.out[cb * 64 + tid] = (float)s;
    }
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(ew + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)p.row_tiles - 1) {  // the last one through: everybody has read this round's epoch long ago
        __hip_atomic_store(ew + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ew, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (round + 1 < p.rounds) {  // several rounds in one launch (timing only): wait for the bump before re-reading it
        unsigned sp = 0;
        while (ld_sc1(ew) != tag) { __builtin_amdgcn_s_sleep(1); if (++sp > kSpinMax) break; }
      }
    }
    __syncthreads();
  }
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

static float* dalloc(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 16777216.0f - 0.5f); }
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.gcnArchName, n_cu);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float* x = dalloc((size_t)kB * kK1, 2.0f, 1);
  float* W1 = dalloc((size_t)kK1 * kN1, 0.2f, 2);
  float* b1 = dalloc(kN1, 0.1f, 3);
  float* W2 = dalloc((size_t)kN1 * kN2, 0.2f, 4);
  float* b2 = dalloc(kN2, 0.1f, 5);
  float* gamma = dalloc(kN1, 1.0f, 6);
  float* beta = dalloc(kN1, 0.1f, 7);
  float* a1 = dalloc((size_t)kB * kN1, 0.f, 8);
  float* h1 = dalloc((size_t)kB * kN1, 0.f, 9);
  float* a2 = dalloc((size_t)kB * kN2, 0.f, 10);
  float* part = dalloc(2 * 32 * 1024, 0.f, 11);
  float* mm = dalloc(2 * kN1, 0.f, 12);
  float* save = dalloc(2 * kN1, 0.f, 13);
  float ms = 0.f;

  // ---- (a) today's three launches
  auto three = [&]() {
    GemmArgs g = gemm_args(x, kK1, W1, kN1, a1, kN1, kB, kN1, kK1, EPI_BIAS_RELU_STATS);
    g.bias = b1; g.part = part;
    launch_gemm(0, 1, g, st);
    BnFwdArgs b;
    memset(&b, 0, sizeof(b));
    b.a = a1; b.h = h1; b.B = kB; b.N = kN1; b.has_bn = 1; b.Bstat = kB; b.part = part; b.n_tiles = stats_row_tiles(kB, kN1, kK1, 0);
    b.gamma = gamma; b.beta = beta; b.mmean = mm; b.mvar = mm + kN1; b.save_mean = save;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((kN1 + 127) / 128, (kB + kApplyRows - 1) / kApplyRows), dim3(256), 0, st, b);
    GemmArgs g2 = gemm_args(h1, kN1, W2, kN2, a2, kN2, kB, kN2, kN1, EPI_BIAS_RELU_STATS);
    g2.bias = b2; g2.part = part;
    launch_gemm(0, 1, g2, st);
  };
  for (int i = 0; i < 20; ++i) three();
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) three();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  const float t_three = ms * 1000.f / reps;
  printf("(a) three launches (gemm 256->128 + stats | bn_apply_fwd | gemm 128->64 + stats): %.2f us per link\n", t_three);
  // the boundary alone: empty kernels
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 3 * reps; ++i) hipLaunchKernelGGL(empty_kernel, dim3(128), dim3(256), 0, st, (int*)nullptr);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("    an empty 128-workgroup launch in the same loop: %.2f us each\n", ms * 1000.f / (3 * reps));

  // ---- (b) persistent teams
  TeamState* ts;
  CK(hipMalloc(&ts, sizeof(TeamState)));
  const size_t lds_b = 84 * 1024;  // > half of the CU's 160 KiB: one workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(team_links_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, team_links_kernel, 256, lds_b));
  printf("(b) team kernel: occupancy query %d workgroup(s) per CU\n", occ);
  LinkArgs la;
  la.x = x; la.W1 = W1; la.b1 = b1; la.W2 = W2; la.b2 = b2; la.gamma = gamma; la.beta = beta; la.a1 = a1; la.h1 = h1; la.a2 = a2;
  la.part = part; la.ts = ts;
  for (int n_wg : {n_cu, n_cu / 2}) {
    la.n_wg = n_wg;
    float t_l[2] = {0.f, 0.f};
    const int Ls[2] = {1, 33};
    unsigned counts[8] = {0};
    unsigned err = 0;
    for (int v = 0; v < 2; ++v) {
      la.links = Ls[v];
      float best = 1e30f;
      for (int rep = 0; rep < 12; ++rep) {
        CK(hipMemsetAsync(ts, 0, sizeof(TeamState), st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(team_links_kernel, dim3(n_wg), dim3(256), lds_b, st, la);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
      }
      t_l[v] = best * 1000.f;
      TeamState h;
      CK(hipMemcpy(&h, ts, sizeof(h), hipMemcpyDeviceToHost));
      memcpy(counts, h.count, sizeof(counts));
      err |= h.err;
    }
    printf("    %3d workgroups: teams by XCC_ID = [%u %u %u %u %u %u %u %u], err %u; 1 link %.2f us, 33 links %.2f us -> %.2f us per link\n",
           n_wg, counts[0], counts[1], counts[2], counts[3], counts[4], counts[5], counts[6], counts[7], err, t_l[0], t_l[1],
           (t_l[1] - t_l[0]) / 32.f);
  }

  // ---- (c) the granule exchange
  const int col_blocks = 4;
  u64* gran;
  unsigned *epoch, *errw;
  float* xout;
  CK(hipMalloc(&gran, (size_t)col_blocks * 32 * 64 * sizeof(u64)));
  CK(hipMemset(gran, 0, (size_t)col_blocks * 32 * 64 * sizeof(u64)));
  CK(hipMalloc(&epoch, col_blocks * 32 * sizeof(unsigned)));
  CK(hipMemset(epoch, 0, col_blocks * 32 * sizeof(unsigned)));
  CK(hipMalloc(&errw, 64));
  CK(hipMemset(errw, 0, 64));
  CK(hipMalloc(&xout, col_blocks * 64 * sizeof(float)));
  for (int cbs : {4, 16}) {
    u64* gr2; unsigned* ep2;
    CK(hipMalloc(&gr2, (size_t)cbs * 32 * 64 * sizeof(u64)));
    CK(hipMemset(gr2, 0, (size_t)cbs * 32 * 64 * sizeof(u64)));
    CK(hipMalloc(&ep2, cbs * 32 * sizeof(unsigned)));
    CK(hipMemset(ep2, 0, cbs * 32 * sizeof(unsigned)));
    float* xo2;
    CK(hipMalloc(&xo2, cbs * 64 * sizeof(float)));
    for (int rts : {16, 32}) {
      float t_r[2];
      const int Rs[2] = {1, 17};
      for (int v = 0; v < 2; ++v) {
        XchgArgs xa;
        xa.gran = gr2; xa.epoch = ep2; xa.err = errw; xa.out = xo2; xa.row_tiles = rts; xa.rounds = Rs[v];
        float best = 1e30f;
        for (int rep = 0; rep < 12; ++rep) {
          CK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(xchg_kernel, dim3(cbs, rts), dim3(256), 0, st, xa);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep >= 2 && ms < best) best = ms;
        }
        t_r[v] = best * 1000.f;
      }
      std::vector<float> ho(cbs * 64);
      CK(hipMemcpy(ho.data(), xo2, ho.size() * sizeof(float), hipMemcpyDeviceToHost));
      unsigned herr = 0;
      CK(hipMemcpy(&herr, errw, 4, hipMemcpyDeviceToHost));
      double want = 0.0;
      for (int t = 0; t < rts; ++t) want += (t + 1) * 0.5 + 5.0;
      printf("(c) exchange, %2d column blocks x %2d row tiles (%d workgroups): 1 round %.2f us, 17 rounds %.2f us -> %.2f us per exchange; "
             "sum check %.1f (want %.1f), err %u\n", cbs, rts, cbs * rts, t_r[0], t_r[1], (t_r[1] - t_r[0]) / 16.f, ho[5], want, herr);
    }
    CK(hipFree(gr2)); CK(hipFree(ep2)); CK(hipFree(xo2));
  }
  return 0;
}
