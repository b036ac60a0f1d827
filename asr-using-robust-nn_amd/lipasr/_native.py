"""ctypes binding of liblipasr.so (include/lipasr.h).

The product has no CPU fallback: if the shared library is missing, importing this module raises.
Device memory, streams and process groups come from PyTorch (plumbing); every arithmetic step on
the hot path goes through the C ABI below.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be
# in the process BEFORE liblipasr.so is opened so that liblipasr's DT_NEEDED libamdhip64.so.7 resolves to
# that same copy: two HIP runtimes in one process cannot both own the device ("no ROCm-capable device").
import torch  # noqa: F401  (plumbing: device memory, streams, process groups)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LIPASR_LIBRARY: another build of the same library (A/B timing of kernel changes); never a different backend
LIB_PATH = os.environ.get("LIPASR_LIBRARY") or os.path.join(_HERE, "liblipasr.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python asr-using-robust-nn_amd/build.py` "
        "(or __graft_entry__.build()).  lipasr has no CPU fallback."
    )

lib = C.CDLL(LIB_PATH)

OK, EINVAL, ENOMEM, EHIP, EUNSUPPORTED, ESTATE = 0, -1, -2, -3, -4, -5
MAX_LAYERS = 16
SEG_W, SEG_B, SEG_GAMMA, SEG_BETA, SEG_MMEAN, SEG_MVAR = range(6)

c_f = C.c_void_p  # device float*
c_h = C.c_void_p
c_s = C.c_void_p
i32 = C.c_int
f32 = C.c_float
u64 = C.c_uint64
sz = C.c_size_t
PI = C.POINTER(C.c_int)
PV = C.POINTER(C.c_void_p)


class DropoutCfg(C.Structure):
    _fields_ = [("mode", C.c_int), ("seed", C.c_uint64), ("step_dev", C.c_void_p), ("masks", PV)]


# name -> (restype, argtypes); mirrors include/lipasr.h line by line
PROTOTYPES = {
    "lipasr_version": (i32, []),
    "lipasr_last_error": (C.c_char_p, []),
    "lipasr_create": (i32, [i32, C.POINTER(c_h)]),
    "lipasr_destroy": (i32, [c_h]),
    "lipasr_timer_create": (i32, [c_h, PI]),
    "lipasr_timer_start": (i32, [c_h, i32, c_s]),
    "lipasr_timer_stop": (i32, [c_h, i32, c_s]),
    "lipasr_timer_elapsed_ms": (i32, [c_h, i32, C.POINTER(f32)]),
    "lipasr_graph_begin": (i32, [c_h, c_s]),
    "lipasr_graph_end": (i32, [c_h, c_s, PI]),
    "lipasr_graph_launch": (i32, [c_h, i32, c_s]),
    "lipasr_stream_create_masked": (i32, [c_h, C.POINTER(C.c_uint32), i32, C.POINTER(c_s)]),
    "lipasr_stream_destroy": (i32, [c_h, c_s]),
    "lipasr_graph_destroy": (i32, [c_h, i32]),
    "lipasr_sigma_max": (i32, [c_h, c_f, i32, i32, c_f, i32, i32, i32, c_f, c_s]),
    "lipasr_project_per_layer": (i32, [c_h, PV, PI, PI, i32, f32, c_f, i32, i32, c_f, c_s]),
    "lipasr_project_product": (i32, [c_h, PV, PI, PI, i32, f32, PI, i32, c_f, c_s]),
    "lipasr_product_norm": (i32, [c_h, PV, PI, PI, i32, c_f, c_s]),
    "lipasr_frobenius_project": (i32, [c_h, c_f, sz, f32, c_s]),
    "lipasr_bn_correction": (i32, [c_h, c_f, c_f, i32, c_f, c_s]),
    "lipasr_sv_clip": (i32, [c_h, c_f, i32, i32, f32, c_f, c_f, c_s]),
    "lipasr_sign_step": (i32, [c_h, c_f, c_f, c_f, sz, f32, f32, c_s]),
    "lipasr_scaler_fit": (i32, [c_h, c_f, i32, i32, c_f, c_f, c_s]),
    "lipasr_scaler_apply": (i32, [c_h, c_f, i32, i32, c_f, c_f, c_f, c_s]),
    "lipasr_gemm_f32": (i32, [c_h, i32, i32, i32, i32, i32, c_f, i32, c_f, i32, c_f, i32, c_s]),
    "lipasr_gemm_f16x2": (i32, [c_h, i32, i32, i32, i32, i32, c_f, i32, c_f, i32, c_f, i32, f32, f32, c_s]),
    "lipasr_mlp_create": (i32, [c_h, i32, PI, PI, C.POINTER(f32), PI, i32, C.POINTER(c_h)]),
    "lipasr_mlp_destroy": (i32, [c_h]),
    "lipasr_mlp_sizes": (i32, [c_h, C.POINTER(sz), C.POINTER(sz)]),
    "lipasr_mlp_segment": (i32, [c_h, i32, i32, C.POINTER(sz), C.POINTER(sz)]),
    "lipasr_mlp_train_fwd_bwd": (i32, [c_h, c_f, c_f, c_f, c_f, i32, f32, C.POINTER(DropoutCfg), c_f, c_f, c_f, c_f, c_s]),
    "lipasr_mlp_train_fwd_bwd_head": (i32, [c_h, c_f, c_f, c_f, c_f, i32, f32, C.POINTER(DropoutCfg), c_f, c_f, c_f, c_f, c_s]),
    "lipasr_mlp_train_dw0": (i32, [c_h, c_f, i32, c_f, c_s]),
    "lipasr_mlp_grad_split": (i32, [c_h, C.POINTER(sz)]),
    "lipasr_mlp_train_segments": (i32, [c_h, PI]),
    "lipasr_mlp_train_segment_exchange": (i32, [c_h, i32, i32, C.POINTER(sz)]),
    "lipasr_mlp_part_floats": (i32, [c_h, C.POINTER(sz)]),
    "lipasr_mlp_train_segment": (i32, [c_h, i32, c_f, c_f, c_f, c_f, i32, f32, C.POINTER(DropoutCfg), c_f, c_f, c_f, c_f, c_f, i32, f32, c_s]),
    "lipasr_mlp_adam_nonneg": (i32, [c_h, c_f, c_f, c_f, c_f, c_f, f32, f32, f32, f32, f32, c_s]),
    "lipasr_mlp_project_product": (i32, [c_h, c_f, f32, PI, i32, c_f, c_s]),
    "lipasr_mlp_adam_project_product": (i32, [c_h, c_f, c_f, c_f, c_f, c_f, f32, f32, f32, f32, f32, f32, PI, i32, c_f, c_s]),
    "lipasr_mlp_project_per_layer": (i32, [c_h, c_f, f32, c_f, i32, i32, c_f, c_s]),
    "lipasr_mlp_product_norm": (i32, [c_h, c_f, c_f, c_s]),
    "lipasr_mlp_predict": (i32, [c_h, c_f, c_f, c_f, i32, c_f, c_f, c_s]),
    "lipasr_mlp_input_grad": (i32, [c_h, c_f, c_f, c_f, c_f, i32, c_f, c_s]),
    "lipasr_mlp_set_compute": (i32, [c_h, i32]),
    "lipasr_mlp_set_gemm_tiles": (i32, [c_h, i32]),
    "lipasr_mlp_set_fuse_bn": (i32, [c_h, i32]),
    "lipasr_mlp_set_cu_budget": (i32, [c_h, i32]),
    "lipasr_mlp_exchange_errors": (i32, [c_h, C.POINTER(C.c_int)]),
    "lipasr_mlp_output_vjp": (i32, [c_h, c_f, c_f, c_f, c_f, i32, i32, c_f, c_f, c_s]),
    "lipasr_mlp_attack_step": (i32, [c_h, c_f, c_f, c_f, c_f, c_f, i32, f32, f32, c_s]),
    "lipasr_mlp_own_labels": (i32, [c_h, c_f, c_f, c_f, i32, c_f, c_s]),
    "lipasr_mfcc_plan": (i32, [c_h, i32, i32, i32]),
    "lipasr_mfcc_plan_ex": (i32, [c_h, i32, i32, i32, i32, i32]),
    "lipasr_mfcc_dims": (i32, [c_h, PI, PI]),
    "lipasr_mfcc_f32": (i32, [c_h, c_f, i32, i32, c_f, c_f, c_f, c_s]),
    "lipasr_resample_f32": (i32, [c_h, c_f, i32, c_f, c_s]),
    "lipasr_mfcc_from_22k": (i32, [c_h, c_f, i32, i32, i32, c_f, c_f, c_f, c_s]),
    "lipasr_mfcc_i16": (i32, [c_h, c_f, c_f, i32, i32, c_f, c_f, c_f, c_s]),
    "lipasr_mfcc_create": (i32, [c_h, i32, i32, i32, i32, i32, C.POINTER(c_h)]),
    "lipasr_mfcc_destroy": (i32, [c_h]),
    "lipasr_mfcc_plan_dims": (i32, [c_h, PI, PI, PI]),
    "lipasr_mfcc_extract": (i32, [c_h, c_f, i32, c_f, i32, i32, c_f, c_f, c_f, c_s]),
    "lipasr_mfcc_plan_resample": (i32, [c_h, c_f, i32, c_f, c_s]),
    "lipasr_mfcc_plan_from_22k": (i32, [c_h, c_f, i32, i32, i32, c_f, c_f, c_f, c_s]),
    "lipasr_mfcc_plan_profile_begin": (i32, [c_h, i32]),
    "lipasr_mfcc_plan_profile_end": (i32, [c_h, C.POINTER(f32), PI]),
    "lipasr_mfcc_plan_set": (i32, [c_h, i32, i32]),
    "lipasr_mfcc_profile_begin": (i32, [c_h, i32]),
    "lipasr_mfcc_profile_end": (i32, [c_h, C.POINTER(f32), PI]),
    "lipasr_add_noise_f32": (i32, [c_h, c_f, i32, i32, i32, f32, f32, u64, c_s]),
    "lipasr_debug_set": (i32, [c_h, i32, i32]),
    "lipasr_debug_gemm_mode": (i32, [i32]),
    "lipasr_debug_launch_count": (C.c_long, [i32]),
    "lipasr_flag_signal": (i32, [c_h, C.c_void_p, i32, c_s]),
    "lipasr_flag_wait": (i32, [c_h, C.c_void_p, i32, i32, C.c_void_p, c_s]),
    "lipasr_debug_chain_head": (i32, [i32]),
    "lipasr_debug_table": (i32, [i32, i32, C.POINTER(f32), i32]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


if os.environ.get("LIPASR_CHAIN_HEAD"):  # A/B timing knob (lipasr_debug_chain_head): same product bit for bit
    lib.lipasr_debug_chain_head(int(os.environ["LIPASR_CHAIN_HEAD"]))
if os.environ.get("LIPASR_GEMM_MODE"):  # A/B timing knob (lipasr_debug_gemm_mode): never a different backend
    lib.lipasr_debug_gemm_mode(int(os.environ["LIPASR_GEMM_MODE"]))


def last_error() -> str:
    return (lib.lipasr_last_error() or b"").decode("utf-8", "replace")


class LipasrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"liblipasr error {code}: {msg}")
        self.code = code


def check(rc: int) -> int:
    """Raise for a negative return code: ValueError for LIPASR_EINVAL, LipasrError otherwise."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == EINVAL:
        raise ValueError(f"liblipasr: {msg}")
    raise LipasrError(rc, msg)


def int_array(values):
    arr = (C.c_int * max(1, len(values)))(*[int(v) for v in values])
    return arr


def float_array(values):
    return (C.c_float * max(1, len(values)))(*[float(v) for v in values])


def ptr_array(ptrs):
    return (C.c_void_p * max(1, len(ptrs)))(*[C.c_void_p(int(p) if p else None) for p in ptrs])


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_handles = {}
# Objects that own native resources (pipelines: graphs + masked streams; models: classifier plans; extractors: MFCC
# plans).  shutdown() closes them newest-first and then destroys the handles, so that no HIP object of ours is left
# for the runtime's static destructors ("exit-time SIGSEGV").  Weak references: the registry never keeps
# an object alive.
_owners = []


def register_owner(obj):
    """obj.close() must be idempotent and must not raise once the handle is gone."""
    import weakref

    _owners.append(weakref.ref(obj))
    if len(_owners) > 256:
        _owners[:] = [r for r in _owners if r() is not None]


# Native destroy calls free device memory (hipFree synchronises), which is illegal while THIS thread captures a HIP graph:
# it invalidates the capture ("operation failed due to a previous error during capture").  Python finalisers run whenever
# the garbage collector decides -- also in the middle of TrainPipeline._capture -- so every destroy goes through
# destroy_or_defer(): during a capture it is parked and run right after the capture ends.
_capture_depth = 0
_deferred = []


def capture_enter():
    global _capture_depth
    _capture_depth += 1


def capture_exit():
    global _capture_depth
    _capture_depth = max(0, _capture_depth - 1)
    if _capture_depth == 0:
        pending, _deferred[:] = list(_deferred), []
        for fn, args in pending:
            fn(*args)


def destroy_or_defer(fn, *args):
    if _capture_depth > 0:
        _deferred.append((fn, args))
    else:
        fn(*args)


class Handle:
    """One lipasr handle per (process, device)."""

    def __init__(self, device: int):
        self.device = device
        h = c_h()
        check(lib.lipasr_create(device, C.byref(h)))
        self.h = h

    @property
    def alive(self):
        return bool(self.h)

    def close(self):
        """Closes every live owner made on this handle first (pipelines, models, extractors, newest first): lipasr_destroy frees
        their native plans, and an owner that still held a plan pointer would turn its next call into a use-after-free instead
        of an error.  Closed owners raise on use."""
        if self.h:
            for ref in reversed(list(_owners)):
                obj = ref()
                if obj is not None and (getattr(obj, "h", None) is self or getattr(obj, "_h", None) is self):
                    try:
                        obj.close()
                    except Exception:
                        pass
            lib.lipasr_destroy(self.h)
            self.h = None


def get_handle(device=None) -> Handle:
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("lipasr needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False and there is no CPU path")
    if device is None:
        device = torch.cuda.current_device()
    h = _handles.get(device)
    if h is None or not h.alive:
        h = _handles[device] = Handle(device)
    return h


def shutdown():
    """Deterministic teardown: pipelines / models / extractors (newest first), then the handles -- graphs, CU-masked
    streams, plans, events, scratch -- all while the HIP runtime is still up.  Registered with atexit AFTER torch was
    imported, so it runs BEFORE torch's and the runtime's own exit handlers; safe to call more than once."""
    global _capture_depth
    _capture_depth = 0
    pending, _deferred[:] = list(_deferred), []
    for fn, args in pending:
        try:
            fn(*args)
        except Exception:
            pass
    owners, _owners[:] = list(_owners), []
    for ref in reversed(owners):
        obj = ref()
        if obj is not None:
            try:
                obj.close()
            except Exception:
                pass
    for h in list(_handles.values()):
        try:
            h.close()
        except Exception:
            pass
    _handles.clear()


import atexit  # noqa: E402

atexit.register(shutdown)


def debug_table(which: int, sr_in: int = 16000):
    """Host-only constant tables as the kernels read them (numpy float32)."""
    import numpy as np

    n = check(lib.lipasr_debug_table(which, sr_in, None, 0))
    out = np.zeros(n, dtype=np.float32)
    check(lib.lipasr_debug_table(which, sr_in, out.ctypes.data_as(C.POINTER(C.c_float)), n))
    return out
