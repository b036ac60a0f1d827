"""Why did the 128 x 64 exchange tile not run inside the full suite?  Launch counts after one float16x2 step on a 128-CU budget,
optionally after other work in the same process (argv: names of warm-ups to run first)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "asr-using-robust-nn_amd"); sys.path.insert(0, ".")
import numpy as np, torch
from helpers import build_model, dev, load_params
from oracle import mlp_ref as P
from lipasr import _native as N

widths = (880, 1024, 512, 256, 128, 64, 10)
spec = [P.LayerSpec(widths[i], widths[i + 1], i + 2 < len(widths), 0.0, True) for i in range(len(widths) - 1)]
p = P.init_params(spec, seed=1, dtype=np.float32, nonneg_init=True)
rng = np.random.default_rng(0)
x = dev(rng.standard_normal((1024, 880)).astype(np.float32)); y = dev(P.to_categorical(rng.integers(0, 10, 1024), 10))
for w in sys.argv[1:]:
    if w == "f32":
        m = build_model(spec, max_batch=1024, compute_dtype="float32"); load_params(m, p); m.train_fwd_bwd(x, y); m.close()
    if w == "f16small":
        m = build_model(spec, max_batch=256, compute_dtype="float16x2"); load_params(m, p); m.train_fwd_bwd(x[:256], y[:256]); m.close()
    if w == "f16":
        m = build_model(spec, max_batch=1024, compute_dtype="float16x2"); load_params(m, p); m.train_fwd_bwd(x, y); m.close()
    if w == "predict":
        m = build_model(spec, max_batch=1024, compute_dtype="float16x2"); load_params(m, p); m.predict(x.cpu().numpy()); m.close()
m = build_model(spec, max_batch=1024, compute_dtype="float16x2"); load_params(m, p)
N.check(N.lib.lipasr_mlp_set_cu_budget(m._plan, 128)); N.check(N.lib.lipasr_mlp_set_gemm_tiles(m._plan, 128))
c0 = (N.lib.lipasr_debug_launch_count(0), N.lib.lipasr_debug_launch_count(1))
m.train_fwd_bwd(x, y); torch.cuda.synchronize()
c1 = (N.lib.lipasr_debug_launch_count(0), N.lib.lipasr_debug_launch_count(1))
print(sys.argv[1:], "ring2 launches", c1[0] - c0[0], "split-pass dW launches", c1[1] - c0[1], "exchange errors", m.exchange_errors())
