"""The end-to-end training step of BASELINE configs 3-5 as one stream-ordered native sequence:

    waveform batch --K1--> MFCC --A2 affine (fused)--> [PGD-k on the features, K2+K4]
        --K2--> fwd/bwd --(RCCL all-reduce)--> K5 Adam+NonNeg --K3--> Lipschitz projection

Everything after the MFCC reads and writes fixed device buffers, so it is captured once per batch size
into HIP graphs (lipasr_graph_*) and replayed; Adam's step count, the dropout counter and all
projection scalars live in device memory, so a replay needs no host value.  The MFCC kernels read the
resident waveform pool in place and are launched eagerly (3 launches per step).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _native as N
from .extract_features_construct_dataset import MfccExtractor
from .parallel import DataParallel


class TrainPipeline:
    def __init__(self, model, batch, sr_in=16000, n_samp=16000, utterance_length=44, rho=0.1, constraint="product",
                 affine=None, pgd=None, dp=None, use_graph=True, per_layer_iters=4):
        """constraint: 'product' (simple_norm_constraint, all layers), 'per_layer' (norm_constraint) or None.
        affine: (mean, scale) float64 device tensors [20*utterance_length] or None.
        pgd: dict(eps=, eps_step=, max_iter=) for adversarial training on the standardised features."""
        self.model, self.batch, self.L = model, int(batch), int(utterance_length)
        self.dev = model._device
        self.h = N.get_handle(self.dev.index)
        self.ex = MfccExtractor(sr_in, n_samp, self.batch, self.dev)
        self.rho, self.constraint, self.pgd = float(rho), constraint, pgd
        self.dp = dp if dp is not None else DataParallel()
        self.use_graph = use_graph
        self.per_layer_iters = per_layer_iters
        self.mean, self.scale = affine if affine is not None else (None, None)
        nf = 20 * self.L
        if model._widths[0] != nf:
            raise ValueError(f"model input width {model._widths[0]} != {nf}")
        self.feats = torch.zeros(self.batch, nf, device=self.dev)
        self.x_adv = torch.zeros(self.batch, nf, device=self.dev) if pgd else None
        self.labels = torch.zeros(self.batch, model._n_classes, device=self.dev)
        nl = len(model._blocks)
        self.norms = torch.zeros(nl + 1, device=self.dev)
        self.sigmas = torch.zeros(nl, device=self.dev)
        self.v_state = torch.zeros(sum(model._widths[1:]), device=self.dev)
        self._order = N.int_array(list(range(nl)))
        self._warm = False
        self.stream = torch.cuda.Stream(device=self.dev)
        self._graphs = {}

    # ---- pieces (all enqueue on the current stream)
    def _attack_and_train(self, bsz):
        m = self.model
        x = self.feats[:bsz]
        y = self.labels[:bsz]
        if self.pgd:
            xa = self.x_adv[:bsz]
            xa.copy_(x)
            for _ in range(int(self.pgd.get("max_iter", 20))):
                N.check(N.lib.lipasr_mlp_attack_step(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xa), N.ptr(x), N.ptr(y), bsz,
                                                     float(self.pgd.get("eps_step", 0.1)), float(self.pgd["eps"]), N.stream_ptr()))
            x = xa
        m.train_fwd_bwd(x, y, inv_batch=1.0 / (bsz * self.dp.world))

    def _update(self):
        m = self.model
        m.apply_adam()
        if self.constraint == "product":
            N.check(N.lib.lipasr_mlp_project_product(m._plan, N.ptr(m._params), self.rho, self._order, len(m._blocks), N.ptr(self.norms),
                                                     N.stream_ptr()))
        elif self.constraint == "per_layer":
            N.check(N.lib.lipasr_mlp_project_per_layer(m._plan, N.ptr(m._params), self.rho, N.ptr(self.v_state), 1, self.per_layer_iters,
                                                       N.ptr(self.sigmas), N.stream_ptr()))

    def _capture(self, fn, *args):
        gid = C.c_int()
        N.check(N.lib.lipasr_graph_begin(self.h.h, N.stream_ptr()))
        try:
            fn(*args)
        finally:
            N.check(N.lib.lipasr_graph_end(self.h.h, N.stream_ptr(), C.byref(gid)))
        return gid.value

    def _warm_start(self):
        if self.constraint == "per_layer" and not self._warm:
            m = self.model
            # cold power iteration once, outside any graph, so the captured step can always run warm
            N.check(N.lib.lipasr_mlp_project_per_layer(m._plan, N.ptr(m._params), self.rho, N.ptr(self.v_state), 0, 48, N.ptr(self.sigmas),
                                                       N.stream_ptr()))
            self._warm = True

    def step(self, waves, y_onehot):
        """waves: float32 device tensor [b, n_samp] (a view into a resident pool is fine), y_onehot [b, classes]."""
        bsz = waves.shape[0]
        with torch.cuda.stream(self.stream):
            self._warm_start()
            self.ex(waves, self.L, self.mean, self.scale, out=self.feats[:bsz])
            self.labels[:bsz].copy_(y_onehot)
            if not self.use_graph:
                self._attack_and_train(bsz)
                self.dp.allreduce_grads(self.model._grads)
                self._update()
                return
            g = self._graphs.get(bsz)
            if g is None:
                # run once eagerly (also warms every kernel), then capture
                if self.dp.world == 1:
                    g = (self._capture(lambda: (self._attack_and_train(bsz), self._update())),)
                else:
                    g = (self._capture(self._attack_and_train, bsz), self._capture(self._update))
                self._graphs[bsz] = g
            N.check(N.lib.lipasr_graph_launch(self.h.h, g[0], N.stream_ptr()))
            if len(g) == 2:
                self.dp.allreduce_grads(self.model._grads)
                N.check(N.lib.lipasr_graph_launch(self.h.h, g[1], N.stream_ptr()))

    def synchronize(self):
        self.stream.synchronize()
