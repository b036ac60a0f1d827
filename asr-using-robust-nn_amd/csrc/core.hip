// Handle lifetime, error reporting, HIP-event timers and HIP-graph capture for liblipasr.
#include "common.h"

namespace lipasr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void mfcc_plan_free(MfccPlan* p);  // mfcc.hip
void mlp_plan_free(lipasr_mlp* m);   // dense.hip

}  // namespace lipasr

using namespace lipasr;

extern "C" {

int lipasr_version(void) { return 500; }  // round 5 (round 4 added lipasr_flag_*, lipasr_debug_chain_head; round 5: see include/lipasr.h)

const char* lipasr_last_error(void) { return g_err; }

int lipasr_create(int device, lipasr_handle_t* out) {
  LP_CHECK_ARG(out != nullptr, "lipasr_create: out is null");
  int n = 0;
  LP_HIP(hipGetDeviceCount(&n));
  LP_CHECK_ARG(device >= 0 && device < n, "lipasr_create: device %d out of range (have %d)", device, n);
  hipDeviceProp_t prop;
  LP_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("lipasr_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return LIPASR_EUNSUPPORTED;
  }
  DeviceGuard g(device);
  if (!g.ok) { set_error("lipasr_create: hipSetDevice(%d) failed", device); return LIPASR_EHIP; }
  lipasr_ctx* c = new lipasr_ctx();
  c->device = device;
  c->scratch_floats = kScratchFloats;
  if (hipMalloc(&c->scratch, c->scratch_floats * sizeof(float)) != hipSuccess) {
    delete c;
    set_error("lipasr_create: scratch allocation failed");
    return LIPASR_ENOMEM;
  }
  if (hipMemset(c->scratch, 0, c->scratch_floats * sizeof(float)) != hipSuccess) {
    (void)hipFree(c->scratch);
    delete c;
    set_error("lipasr_create: scratch memset failed");
    return LIPASR_EHIP;
  }
  if (hipMalloc(&c->zeros, 256) != hipSuccess || hipMemset(c->zeros, 0, 256) != hipSuccess) {
    (void)hipGetLastError();
    if (c->zeros) (void)hipFree(c->zeros);
    c->zeros = nullptr;  // (the ring kernel then takes only K that are multiples of 32)
  }
  *out = c;
  return LIPASR_OK;
}

// Everything the handle made is released here, in dependency order, while the HIP runtime is still up: a HIP object
// that outlives the process's exit handlers is torn down by the runtime's own static destructors, and for a CU-masked
// stream (a hardware queue of its own) with graph executables instantiated on it that teardown ran after the
// profiler's queue interception had been finalised ("exit-time SIGSEGV").  Order: drain the device, graph
// executables (they reference streams' captured nodes and plan buffers), masked streams, plans, events, scratch.
int lipasr_destroy(lipasr_handle_t h) {
  LP_CHECK_ARG(h != nullptr, "lipasr_destroy: null handle");
  DeviceGuard g(h->device);
  (void)hipDeviceSynchronize();
  for (hipGraphExec_t ge : h->graphs)
    if (ge) (void)hipGraphExecDestroy(ge);
  h->graphs.clear();
  for (hipStream_t st : h->streams)
    if (st) (void)hipStreamDestroy(st);
  h->streams.clear();
  // plans still alive are the handle's to free (their pointers die with it)
  std::vector<lipasr_mlp*> mlps;
  mlps.swap(h->mlps);
  for (lipasr_mlp* m : mlps) mlp_plan_free(m);
  std::vector<MfccPlan*> plans;
  plans.swap(h->mfcc_plans);
  for (MfccPlan* p : plans) mfcc_plan_free(p);
  h->mfcc = nullptr;
  for (hipEvent_t e : h->timers) (void)hipEventDestroy(e);
  h->timers.clear();
  if (h->scratch) (void)hipFree(h->scratch);
  if (h->zeros) (void)hipFree(h->zeros);
  delete h;
  return LIPASR_OK;
}

// ------------------------------------------------------------------ timers
int lipasr_timer_create(lipasr_handle_t h, int* timer_id) {
  LP_CHECK_ARG(h && timer_id, "lipasr_timer_create: null argument");
  DeviceGuard g(h->device);
  hipEvent_t a, b;
  LP_HIP(hipEventCreate(&a));
  LP_HIP(hipEventCreate(&b));
  *timer_id = (int)(h->timers.size() / 2);
  h->timers.push_back(a);
  h->timers.push_back(b);
  return LIPASR_OK;
}

int lipasr_timer_start(lipasr_handle_t h, int id, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && id >= 0 && (size_t)(2 * id + 1) < h->timers.size(), "lipasr_timer_start: bad timer id %d", id);
  LP_HIP(hipEventRecord(h->timers[2 * id], S(stream)));
  return LIPASR_OK;
}

int lipasr_timer_stop(lipasr_handle_t h, int id, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && id >= 0 && (size_t)(2 * id + 1) < h->timers.size(), "lipasr_timer_stop: bad timer id %d", id);
  LP_HIP(hipEventRecord(h->timers[2 * id + 1], S(stream)));
  return LIPASR_OK;
}

int lipasr_timer_elapsed_ms(lipasr_handle_t h, int id, float* ms_host) {
  LP_CHECK_ARG(h && ms_host && id >= 0 && (size_t)(2 * id + 1) < h->timers.size(),
               "lipasr_timer_elapsed_ms: bad argument");
  LP_HIP(hipEventSynchronize(h->timers[2 * id + 1]));
  LP_HIP(hipEventElapsedTime(ms_host, h->timers[2 * id], h->timers[2 * id + 1]));
  return LIPASR_OK;
}

// ------------------------------------------------------------------ device-side ordering between two streams
// A counter in device memory that one stream raises and another waits for, both as one-wavefront kernels: the dependency never
// passes through the host or the command processor's event machinery (a hipEventRecord + hipStreamWaitEvent pair per hand-off
// cost the pipeline 9 us per step).  A wait REPORTS after timeout_ms (*err = 1; `err` may be pinned host memory, so
// the host sees it without synchronising anything) and KEEPS WAITING -- round 4 let the stream go on at that point, i.e. onto
// data the other stream had not finished: an ordering guarantee must not turn into a best effort because a peer sat
// in a long collective or a serialising profiler ran.  Only after kFlagHardFactor x timeout_ms does it give up (*err = 2), so that
// a signal that can never come (a bug, a destroyed stream) still drains the queue instead of hanging the GPU; and a wait that
// finds *err already at 2 leaves at once: after one abandoned wait nothing later on that pipeline is ordered, the host is
// expected to have raised long before.  The two streams must be able to run at the same time (disjoint CU masks, or spare
// wave slots): the waiting wavefront occupies one slot.
namespace lipasr {
constexpr long long kFlagHardFactor = 4;
__global__ void flag_signal_kernel(int* flag, int value) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void flag_wait_kernel(const int* flag, int value, int* err, long long soft_ticks) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();  // 100 MHz, constant
    bool reported = false;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
      __builtin_amdgcn_s_sleep(16);
      const long long dt = wall_clock64() - t0;
      if (!reported && dt > soft_ticks) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= 2) break;  // an earlier wait was abandoned
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        reported = true;
      }
      if (dt > kFlagHardFactor * soft_ticks) { __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
  }
}
}  // namespace lipasr

int lipasr_flag_signal(lipasr_handle_t h, int* flag, int value, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && flag, "lipasr_flag_signal: null argument");
  hipLaunchKernelGGL(lipasr::flag_signal_kernel, dim3(1), dim3(64), 0, S(stream), flag, value);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

int lipasr_flag_wait(lipasr_handle_t h, const int* flag, int value, int timeout_ms, int* err, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && flag && err, "lipasr_flag_wait: null argument");
  LP_CHECK_ARG(timeout_ms >= 1 && timeout_ms <= 150000, "lipasr_flag_wait: timeout %d ms outside [1, 150000]", timeout_ms);
  hipLaunchKernelGGL(lipasr::flag_wait_kernel, dim3(1), dim3(64), 0, S(stream), flag, value, err, (long long)timeout_ms * 100000LL);
  LP_LAUNCH_CHECK();
  return LIPASR_OK;
}

// ------------------------------------------------------------------ graphs
int lipasr_graph_begin(lipasr_handle_t h, lipasr_stream_t stream) {
  LP_CHECK_ARG(h != nullptr, "lipasr_graph_begin: null handle");
  LP_HIP(hipStreamBeginCapture(S(stream), hipStreamCaptureModeThreadLocal));
  return LIPASR_OK;
}

int lipasr_graph_end(lipasr_handle_t h, lipasr_stream_t stream, int* graph_id) {
  LP_CHECK_ARG(h && graph_id, "lipasr_graph_end: null argument");
  hipGraph_t graph = nullptr;
  LP_HIP(hipStreamEndCapture(S(stream), &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return LIPASR_EHIP;
  }
  *graph_id = (int)h->graphs.size();
  h->graphs.push_back(exec);
  return LIPASR_OK;
}

int lipasr_graph_launch(lipasr_handle_t h, int id, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && id >= 0 && (size_t)id < h->graphs.size() && h->graphs[id], "lipasr_graph_launch: bad graph id %d", id);
  LP_HIP(hipGraphLaunch(h->graphs[id], S(stream)));
  return LIPASR_OK;
}

int lipasr_graph_destroy(lipasr_handle_t h, int id) {
  LP_CHECK_ARG(h && id >= 0 && (size_t)id < h->graphs.size() && h->graphs[id], "lipasr_graph_destroy: bad graph id %d", id);
  LP_HIP(hipGraphExecDestroy(h->graphs[id]));
  h->graphs[id] = nullptr;
  return LIPASR_OK;
}

// ------------------------------------------------------------------ CU-masked streams
int lipasr_stream_create_masked(lipasr_handle_t h, const uint32_t* cu_mask, int n_words, lipasr_stream_t* out) {
  LP_CHECK_ARG(h && cu_mask && out && n_words >= 1 && n_words <= 64, "lipasr_stream_create_masked: bad argument");
  bool any = false;
  for (int i = 0; i < n_words; ++i) any = any || cu_mask[i] != 0;
  LP_CHECK_ARG(any, "lipasr_stream_create_masked: the mask enables no CU");
  DeviceGuard g(h->device);
  hipStream_t st = nullptr;
  LP_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask));
  h->streams.push_back(st);
  *out = st;
  return LIPASR_OK;
}

int lipasr_stream_destroy(lipasr_handle_t h, lipasr_stream_t stream) {
  LP_CHECK_ARG(h && stream, "lipasr_stream_destroy: null argument");
  DeviceGuard g(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  bool mine = false;
  for (hipStream_t s : h->streams) mine = mine || s == st;
  LP_CHECK_ARG(mine, "lipasr_stream_destroy: not a live stream of this handle");
  // the registry slot is cleared only once the queue is really gone: if either call fails the stream stays registered and
  // lipasr_destroy retries it, instead of leaving a hardware queue to the runtime's static destructors
  LP_HIP(hipStreamSynchronize(st));
  LP_HIP(hipStreamDestroy(st));
  for (hipStream_t& s : h->streams)
    if (s == st) s = nullptr;
  return LIPASR_OK;
}

}  // extern "C"
