"""The end-to-end training step of BASELINE configs 3-5 as one stream-ordered native sequence:

    waveform batch --K1--> MFCC --A2 affine (fused)--> [PGD-k on the features, K2+K4]
        --K2--> fwd/bwd --(RCCL all-reduce)--> K5 Adam+NonNeg --K3--> Lipschitz projection

Everything after the MFCC reads and writes fixed device buffers, so it is captured once per batch size
into HIP graphs (lipasr_graph_*) and replayed; Adam's step count, the dropout counter and all
projection scalars live in device memory, so a replay needs no host value.  The MFCC kernels read the
resident waveform pool in place and are launched eagerly (3 launches per step).

Two HIP streams: the MFCC of batch i+1 (VALU/LDS-bound kernels) runs on its own stream while the
classifier step of batch i (many short MFMA / latency-bound kernels) runs on the training stream; the
feature and label buffers are double-buffered and hipEvents carry the two dependencies
(features ready -> train; train done with a buffer -> MFCC may overwrite it).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N
from .extract_features_construct_dataset import MfccExtractor
from .parallel import DataParallel


class TrainPipeline:
    def __init__(self, model, batch, sr_in=16000, n_samp=16000, utterance_length=44, rho=0.1, constraint="product",
                 affine=None, pgd=None, dp=None, use_graph="auto", per_layer_iters=4, extractor=None, mfcc_cus="auto",
                 sync_inputs=True, overlap_buckets=False, sync_bn=False, train_cus="auto"):
        """constraint: 'product' (simple_norm_constraint, all layers), 'per_layer' (norm_constraint) or None.
        affine: (mean, scale) float64 device tensors [20*utterance_length] or None.
        pgd: dict(eps=, eps_step=, max_iter=) for adversarial training on the standardised features.
        extractor: a feature extractor ``f(waves, mean, scale, out=)`` replacing the 2048/512 MFCC plan, e.g.
        ``speaker_recognition.WindowMfcc`` (441/220 windows -> 2020 features; pass utterance_length=101).
        mfcc_cus: how many of the GPU's CUs the feature-extraction stream may use (a CU-masked HIP stream, spread
        evenly over the XCDs); "auto" = the measured best share, None / 0 = no mask.
        train_cus: "rest" confines the classifier's stream to the CUs the extraction stream does NOT use (a second CU-masked
        stream: the chain's short kernels then never share a CU's LDS, wave slots and L1 with an MFCC workgroup), "all"
        leaves it on every CU, "auto" = "rest" with the built-in MFCC plan.
        use_graph: replay the classifier's part of a step as HIP graph(s) (True), launch its kernels one by one (False), or
        "auto": graphs for the long launch sequences (PGD adversarial training: ~440 launches per step; synchronized BatchNorm:
        one graph per segment between collectives), plain launches for the 29-kernel step -- measured at the end of round 4
        (batch 1024, MI355X): 0.411 against 0.417 ms per step over 200 steps, 0.433 against 0.444 over the driver's 20 (config 2:
        0.366 against 0.373).  The host needs 0.25 ms to enqueue a step's launches (0.09 with the graph), so it stays ahead of
        the GPU either way, and a replayed graph costs the GPU more between nodes than back-to-back launches on one queue do
        (the same graph under AMD_DIRECT_DISPATCH=0: 0.345 ms in config 2 -- but the two-stream pipeline 0.82).
        sync_inputs: order the extraction stream after the caller's current stream and the caller's stream after the
        extraction (two event hand-offs per step, ~50 us of a 0.4 ms step).  Needed whenever the tensors handed to step()
        were just produced on the caller's stream or are temporaries; a loop over a resident pool that was filled and
        synchronised beforehand (bench.py) may pass False."""
        self.sync_inputs = bool(sync_inputs)
        self._train_cus = train_cus
        # Data parallel, gradient exchange.  False: two HIP graphs around ONE all-reduce of the whole flat buffer.  True:
        # three graphs around two buckets, everything but [dW_0 | db_0] (44 % of the bytes) reduced beside the dW_0 GEMM.
        # Measured on one MI355X with a one-rank RCCL group (identity collectives, scratch/nccl_one_rank.py): the extra
        # graph boundary and the second collective's stream hand-offs cost +57 us per step against +14 us for the
        # one-collective schedule, more than the ~36 us of wire time the overlap can hide at this message size.
        self.overlap_buckets = bool(overlap_buckets)
        # Data parallel, BatchNorm: per-replica statistics (default; accuracy parity shown in tests/test_dp_gpu.py) or, opt-in,
        # statistics of the global batch through 2 small all-reduces per BatchNorm layer (eager launches, no HIP graph): set below
        self.model, self.batch, self.L = model, int(batch), int(utterance_length)
        self.dev = model._device
        self.h = N.get_handle(self.dev.index)
        self.ex = MfccExtractor(sr_in, n_samp, self.batch, self.dev) if extractor is None else extractor
        self._custom_ex = extractor is not None
        self.rho, self.constraint, self.pgd = float(rho), constraint, pgd
        self.dp = dp if dp is not None else DataParallel()
        model._replica_rank = self.dp.rank  # every rank draws its own dropout masks
        self._late = model.late_floats      # [dW_0 | db_0]: the gradient bucket that is ready last
        self.sync_bn = bool(sync_bn) and self.dp.world > 1
        self.use_graph = (bool(pgd) or self.sync_bn) if use_graph == "auto" else bool(use_graph)
        self.per_layer_iters = per_layer_iters
        self.mean, self.scale = affine if affine is not None else (None, None)
        nf = 20 * self.L
        if model._widths[0] != nf:
            raise ValueError(f"model input width {model._widths[0]} != {nf}")
        self._nbuf = 2  # feature/label buffers in flight (a third or fourth measured no different, CU-masked or not, balanced legs or not)
        self._feats2 = [torch.zeros(self.batch, nf, device=self.dev) for _ in range(self._nbuf)]
        self._labels2 = [torch.zeros(self.batch, model._n_classes, device=self.dev) for _ in range(self._nbuf)]
        self.feats, self.labels = self._feats2[0], self._labels2[0]  # buffers of the most recent step
        self.x_adv = torch.zeros(self.batch, nf, device=self.dev) if pgd else None
        nl = len(model._blocks)
        self.norms = torch.zeros(nl + 1, device=self.dev)
        self.sigmas = torch.zeros(nl, device=self.dev)
        self.v_state = torch.zeros(sum(model._widths[1:]), device=self.dev)
        self._order = N.int_array(list(range(nl)))
        self._warm = False
        self._masked_stream = self._masked_train_stream = None  # CU-masked streams this pipeline made (hardware queues of their own)
        self._gemm_tiles_set = False
        self.stream = torch.cuda.Stream(device=self.dev)       # training stream (equal priorities measured best)
        self.mfcc_stream = self._make_mfcc_stream(mfcc_cus)   # feature-extraction stream
        # When the classifier's stream is IDLE (the first step after a drained pipeline, a host-bound caller), the extraction does
        # not need to keep to its CU share: it runs on an unmasked stream, 0.19 instead of 0.39 ms for 1024 clips -- in a 20-step
        # timed region that first extraction is 0.02 ms per step (round 4, scratch/fill_probe.py).  LIPASR_WIDE_WHEN_IDLE=0: off.
        import os as _os

        self._wide_stream = torch.cuda.Stream(device=self.dev) if (self._masked_stream is not None and not self._custom_ex
                                                                  and _os.environ.get("LIPASR_WIDE_WHEN_IDLE", "1") == "1") else None
        self._ev_last_mfcc = None     # the extraction plan's scratch is shared: consecutive extractions are ordered, whatever stream they ran on
        self._last_xs = None
        self._last_train_ev = None
        self._n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        # The two hand-offs between the streams (features of buffer b ready -> classifier; classifier done with buffer b -> next
        # extraction into it) are device-side counters (lipasr_flag_signal / lipasr_flag_wait: one-wavefront kernels) instead of
        # hipEventRecord + hipStreamWaitEvent pairs: 0.405 against 0.409 ms per step (round 4, batch 1024, same box back to back;
        # config 5 3.84 against 4.05-4.09 ms).  Needs streams that can run at the same time whatever the other one holds: the two
        # hardware queues of _make_mfcc_stream.  Round 5: a wait that times out REPORTS and keeps
        # waiting (it used to let the stream run on onto unordered data); the report lands in a pinned host word that step(),
        # synchronize() and close() read without synchronising anything, so a caller that never calls pipe.synchronize() hears of
        # it at its next step; and the default is events whenever ranks exchange gradients or BatchNorm sums (a rank sitting in a
        # collective while a peer checkpoints is a legitimate stall, and the flag kernels have never run beside an RCCL kernel:
        # no N > 1 run exists).  LIPASR_GPU_FLAGS=1 forces the flags on, 0 forces events (needed under tools that serialise
        # kernels across streams, e.g. rocprofv3 --pmc: a wait kernel would spin in front of the signal it waits for).
        self._flags = None
        self._flag_err = None
        want_flags = _os.environ.get("LIPASR_GPU_FLAGS", "auto")
        if want_flags == "auto":
            want_flags = "1" if (self.dp.world == 1 and not self.sync_bn) else "0"
        if want_flags == "1" and self._masked_train_stream is not None:
            self._flags = torch.zeros(2 * self._nbuf, dtype=torch.int32, device=self.dev)  # ready[b] | free[b]
            self._flag_err = torch.zeros(1, dtype=torch.int32).pin_memory()                # 1: a wait is overdue, 2: a wait gave up
            self._flag_timeout_ms = int(_os.environ.get("LIPASR_FLAG_TIMEOUT_MS", "30000"))  # what a wait sits out before it reports
        self._ev_feat = [torch.cuda.Event() for _ in range(self._nbuf)]   # features of buffer b are ready
        self._ev_free = [None] * self._nbuf                             # training has finished reading buffer b
        self._i = 0
        self._graphs = {}
        self._closed = False
        self._prof = None
        N.register_owner(self)

    def _make_mfcc_stream(self, mfcc_cus):
        """The MFCC kernels are throughput kernels with thousands of workgroups; the classifier step is a chain of ~30
        short dependent kernels.  Sharing every CU, the chain's workgroups queue behind MFCC workgroups for LDS and
        wave slots and the step stretches.  A CU-masked stream keeps the MFCC work on part of the chip: it runs
        slower there, but it is off the critical path, and the chain finds the other CUs free."""
        import os

        n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        if mfcc_cus == "auto":
            # measured at per-GPU batch 1024 on MI355X: 0.591 ms per step unmasked, 0.549 / 0.529 / 0.518 / 0.513 / 0.510 /
            # 0.506 / 0.532 / 0.528 with 240 / 224 / 208 / 192 / 176 / 160 / 144 / 128 CUs for the MFCC stream
            # (a heavier extractor wants a larger share: the 441/220 Speaker-recognition path measured 0.904 ms unmasked,
            # 0.865 / 0.826 / 0.811 with 160 / 192 / 224 CUs)
            # re-measured at the end of round 2 (after the STFT / GEMM changes): batch 1024: 0.540 / 0.505 / 0.489 / 0.520 ms
            # with 128 / 160 / 192 / 224 CUs; batch 512: 0.340 with 160, 0.349 with 192 -- the MFCC work grows with the
            # batch, the classifier's kernel chain hardly does, so the larger batch wants the larger share
            env = os.environ.get("LIPASR_MFCC_CUS")
            if env is not None:
                mfcc_cus = int(env)
            elif self._custom_ex:
                mfcc_cus = (n_cu * 7) // 8
            else:
                # round 3 (resampler on the fp16 matrix instruction, dual-FFT STFT kernel: the MFCC needs 155 us of the
                # whole chip instead of 240), 200-step runs, classifier stream on every CU: batch 1024: 0.499 / 0.460 / 0.481 /
                # 0.493 / 0.499 ms with 96 / 128 / 160 / 192 / 224 CUs; batch 512: 0.341 with 128, 0.353 with 160.
                # With the classifier's stream on the REST of the chip (train_cus="rest", 300-step runs, three repeats within
                # 0.002): batch 1024: 0.616 / 0.447 / 0.471 ms with 64 / 96 / 128 CUs for the MFCC (0.468 for the best shared
                # schedule, 0.517 with 96 CUs and the classifier everywhere); batch 512: 0.339 / 0.346 with 64 / 96 (0.347 shared)
                # PGD adversarial training is the other regime: the classifier's leg is ten times the MFCC's and throughput-bound
                # on its GEMMs, so it keeps every CU (PGD-20, batch 1024: 4.09 ms with 128 CUs for the MFCC and the classifier
                # everywhere; 4.14 / 4.25 with 64 / 32; on the rest of the chip only: 4.67 / 5.05 / 5.24 with 32 / 64 / 96)
                rest = self._train_cus == "rest" or (self._train_cus == "auto" and not self.pgd
                                                     and os.environ.get("LIPASR_TRAIN_CUS", "rest") == "rest")
                if rest:
                    # after the grouped dW GEMM moved to LDS tiles: batch 1024: 0.614 / 0.431 / 0.453 with 64 / 96 / 128;
                    # batch 2048: 0.871 / 0.679 with 96 / 128; batch 512: 0.337 with 64
                    mfcc_cus = n_cu // 4 if self.batch <= 768 else ((n_cu * 3) // 8 if self.batch <= 1536 else n_cu // 2)
                    # round 5: with the classifier's training GEMMs in arithmetic mode 2 (fp16 two-plane split, LDS-DMA ring) its
                    # leg is 0.36 ms on 160 CUs and 0.37-0.38 on 128, so the extraction gets half the chip: batch 1024
                    # 0.371 ms per step with 128 | 128 against 0.400 with 96 | 160 (200 steps, same box, interleaved)
                    if getattr(self.model, "_compute_dtype", "float32") == "float16x2" and 768 < self.batch <= 1536:
                        mfcc_cus = n_cu // 2
                else:
                    mfcc_cus = n_cu // 2
        if not mfcc_cus or mfcc_cus >= n_cu:
            # no partition -- but a long dependent chain (the PGD graph: ~440 nodes) still must not sit on a pool stream that may
            # share a hardware queue with the extraction's pool stream (the cause of round 3's 3x; ADVICE r4: this fix used to
            # be reachable only through the masked-MFCC branch below)
            if self.pgd:
                self._own_queue_train_stream(n_cu)
            return torch.cuda.Stream(device=self.dev)
        # Measured on MI355X (scratch/cu_mask_probe.py): mask bits act in groups of 8 consecutive bits -- group g
        # (bits 8g .. 8g+7) stands for CU g of every XCD, and the group is enabled when any of its bits is set.  So the
        # share is granted in steps of 8 CUs (one per XCD).
        # (pairs of settings 32 apart behaved alike -- 240/224, 208/192, 176/160, 144/128 -- so the share is rounded
        # down to a multiple of 32 CUs, which is what the hardware appears to grant; re-checked in round 3 with the two
        # disjoint partitions: 13, 14 and 15 groups leave the STFT kernel's time where 12 groups put it, 0.267 ms)
        n_groups = max(1, n_cu // 8)
        k = max(4, min((int(mfcc_cus) // 32) * 4, n_groups))
        words = (n_cu + 31) // 32
        mask = (C.c_uint32 * words)()
        # the first k groups: an evenly spread choice of groups measured erratic (0.517 ... 0.78 ms), the prefix smooth
        for g in range(k):
            for b in range(8 * g, 8 * g + 8):
                mask[b // 32] |= 1 << (b % 32)
        self.mfcc_cus = 8 * k
        st = N.c_s()
        rc = N.lib.lipasr_stream_create_masked(self.h.h, mask, words, C.byref(st))
        if rc != N.OK:  # a runtime without CU masking: same results, the shared-CU schedule
            import warnings

            warnings.warn(f"lipasr: CU-masked stream unavailable ({N.last_error()}); the MFCC stream shares every CU")
            if self.pgd:
                self._own_queue_train_stream(n_cu)
            return torch.cuda.Stream(device=self.dev)
        self._masked_stream = st
        # the three-kernel path's persistent resampler sizes its grid to one workgroup per CU it may use
        if hasattr(self.ex, "set"):
            self.ex.set(1, self.mfcc_cus)
            # (the STFT kernel's fused DCT epilogue, plan key 4, stays off: on a CU share one workgroup per clip holding its 74 kB of
            # LDS while two of its four wavefronts run the fp32 DCT loses to the separate dct_kernel -- stage 0.400 against 0.388 ms
            # on 96 CUs, step 0.434 against 0.418 ms, round 4)
        # the classifier's stream on the CUs the MFCC stream does not use
        want = self._train_cus if self._train_cus != "auto" else os.environ.get("LIPASR_TRAIN_CUS", "all" if (self._custom_ex or self.pgd) else "rest")
        if want == "rest" and k < n_groups:
            mask2 = (C.c_uint32 * words)()
            for g in range(k, n_groups):  # (partitions that share 32 or 64 CUs measured worse: 0.452-0.461 against 0.418 ms)
                for b in range(8 * g, 8 * g + 8):
                    mask2[b // 32] |= 1 << (b % 32)
            st2 = N.c_s()
            if N.lib.lipasr_stream_create_masked(self.h.h, mask2, words, C.byref(st2)) == N.OK:
                self._masked_train_stream = st2
                self.stream = torch.cuda.ExternalStream(st2.value, device=self.dev)
                self.train_cus = 8 * (n_groups - k)
                # on part of the chip the LDS-tiled GEMM pays from fewer tiles on (layer 2 forward, the dX GEMM into layer 2:
                # 128 tiles): -16 us on the classifier's graph at 160 CUs (PGD, which keeps every CU, loses 8 % with it)
                N.check(N.lib.lipasr_mlp_set_gemm_tiles(self.model._plan, int(os.environ.get("LIPASR_GEMM_TILES", "128"))))
                # the fused BatchNorm exchange spins until a column block's workgroups are all resident: tell the plan how many
                # CUs its stream really has (round 5)
                N.check(N.lib.lipasr_mlp_set_cu_budget(self.model._plan, self.train_cus))
                self._gemm_tiles_set = True  # the pipeline owns these settings: close() puts the model's plan back
        else:
            self._own_queue_train_stream(n_cu)
        return torch.cuda.ExternalStream(st.value, device=self.dev)

    def _own_queue_train_stream(self, n_cu):
        """The classifier keeps every CU (PGD, custom extractor, train_cus="all") -- but NOT on a stream of torch's pool: pool
        streams are multiplexed over GPU_MAX_HW_QUEUES (4) hardware queues, and once enough streams exist in the process the
        training stream can land on the queue that carries the MFCC stream's kernels; its 440-node PGD graph then waited
        behind them node by node: 13.0 ms per step instead of 4.1 as the fifth configuration of one process (round 3's
        unexplained 3x; round 4: scratch/pgd_fifth_probe.py, gone with GPU_MAX_HW_QUEUES=8).  A stream made with a FULL CU
        mask owns a hardware queue like the two partition streams do."""
        import os

        if os.environ.get("LIPASR_TRAIN_OWN_QUEUE", "1") != "1":
            return
        words = (n_cu + 31) // 32
        mask2 = (C.c_uint32 * words)()
        for b in range(n_cu):
            mask2[b // 32] |= 1 << (b % 32)
        st2 = N.c_s()
        if N.lib.lipasr_stream_create_masked(self.h.h, mask2, words, C.byref(st2)) == N.OK:
            self._masked_train_stream = st2
            self.stream = torch.cuda.ExternalStream(st2.value, device=self.dev)

    def _check_flags(self):
        """Reads the pinned report word of the device-side waits (no synchronisation).  Raises once per report and clears it,
        so that close() can still drain and release the streams afterwards."""
        fe = self._flag_err
        if fe is None:
            return
        code = int(fe[0])
        if code:
            fe[0] = 0
            self._failed = True
            what = ("is overdue (it is still waiting: nothing ran out of order)" if code == 1
                    else "gave up: what its stream did afterwards was not ordered against the other stream")
            raise RuntimeError(f"TrainPipeline: a device-side wait between the extraction and the classifier stream {what} "
                               f"after {self._flag_timeout_ms} ms without the other stream's signal; LIPASR_GPU_FLAGS=0 uses events")

    # ---- pieces (all enqueue on the current stream)
    def _attack_and_train(self, bsz, b, global_batch, defer_dw0=False):
        m = self.model
        x = self._feats2[b][:bsz]
        y = self._labels2[b][:bsz]
        if self.pgd:
            xa = self.x_adv[:bsz]
            xa.copy_(x)
            for _ in range(int(self.pgd.get("max_iter", 20))):
                N.check(N.lib.lipasr_mlp_attack_step(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xa), N.ptr(x), N.ptr(y), bsz,
                                                     float(self.pgd.get("eps_step", 0.1)), float(self.pgd["eps"]), N.stream_ptr()))
            x = xa
        m.train_fwd_bwd(x, y, inv_batch=1.0 / float(global_batch), defer_dw0=defer_dw0)

    def _attack_only(self, bsz, b):
        """The PGD inner loop alone (inference-mode BatchNorm: nothing to synchronise); returns the training input."""
        m = self.model
        x = self._feats2[b][:bsz]
        if not self.pgd:
            return x
        y = self._labels2[b][:bsz]
        xa = self.x_adv[:bsz]
        xa.copy_(x)
        for _ in range(int(self.pgd.get("max_iter", 20))):
            N.check(N.lib.lipasr_mlp_attack_step(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xa), N.ptr(x), N.ptr(y), bsz,
                                                 float(self.pgd.get("eps_step", 0.1)), float(self.pgd["eps"]), N.stream_ptr()))
        return xa

    def _syncbn_step(self, bsz, b, gb):
        """Synchronized BatchNorm: [PGD loop] -> segment 0 | all-reduce | segment 1 | ... | gradient all-reduce | update.
        With graphs every piece between two collectives is one replayed HIP graph (the collectives themselves are eager
        torch.distributed calls on the stream)."""
        m = self.model
        n_seg = m.syncbn_segments()
        y = self._labels2[b][:bsz]
        xin = (self.x_adv if self.pgd else self._feats2[b])[:bsz]
        if not self.use_graph:
            self._attack_only(bsz, b)
            for seg in range(n_seg):
                n = m.syncbn_segment(seg, xin, y, gb, self.dp.world)
                if n:
                    self.dp.allreduce_grads(m._part[:n])
            self.dp.allreduce_grads(m._grads)
            self._update()
            return
        key = (bsz, b, gb, "syncbn")
        g = self._graphs.get(key)
        if g is None:
            counts = []

            def first():
                self._attack_only(bsz, b)
                counts.append(m.syncbn_segment(0, xin, y, gb, self.dp.world))

            ids = [self._capture(first)]
            for seg in range(1, n_seg):
                ids.append(self._capture(lambda s=seg: counts.append(m.syncbn_segment(s, xin, y, gb, self.dp.world))))
            ids.append(self._capture(self._update))
            self._sync_counts = getattr(self, "_sync_counts", {})
            self._sync_counts[key] = counts
            g = tuple(ids)
            self._graphs[key] = g
        counts = self._sync_counts[key]
        for seg in range(n_seg):
            N.check(N.lib.lipasr_graph_launch(self.h.h, g[seg], N.stream_ptr()))
            if counts[seg]:
                self.dp.allreduce_grads(m._part[:counts[seg]])
        self.dp.allreduce_grads(m._grads)
        N.check(N.lib.lipasr_graph_launch(self.h.h, g[n_seg], N.stream_ptr()))

    def _dw0(self, bsz, b):
        self.model.train_dw0((self.x_adv if self.pgd else self._feats2[b])[:bsz])

    def _reduce_and_update_eager(self, bsz, b):
        """world > 1, no graphs: head -> [bucket A reduces] || dW_0 -> bucket B -> Adam + projection."""
        m = self.model
        late = self._late
        ha = self.dp.allreduce_async(m._grads[late:])
        self._dw0(bsz, b)
        hb = self.dp.allreduce_async(m._grads[:late])
        ha.wait()
        hb.wait()
        self._update()

    def _update(self):
        m = self.model
        if self.constraint == "product":
            # Adam + NonNeg + simple_norm_constraint in one native call (the step counter moves inside the projection)
            m.apply_adam_project_product(self.rho, self._order, self.norms)
            return
        m.apply_adam()
        if self.constraint == "per_layer":
            N.check(N.lib.lipasr_mlp_project_per_layer(m._plan, N.ptr(m._params), self.rho, N.ptr(self.v_state), 1, self.per_layer_iters,
                                                       N.ptr(self.sigmas), N.stream_ptr()))

    def _capture(self, fn, *args):
        gid = C.c_int()
        N.check(N.lib.lipasr_graph_begin(self.h.h, N.stream_ptr()))
        N.capture_enter()  # native destroys (finalisers the garbage collector may run now) wait until the capture is over
        try:
            fn(*args)
        finally:
            try:
                N.check(N.lib.lipasr_graph_end(self.h.h, N.stream_ptr(), C.byref(gid)))
            finally:
                N.capture_exit()
        return gid.value

    def _warm_start(self):
        if self.constraint == "per_layer" and not self._warm:
            m = self.model
            # cold power iteration once, outside any graph, so the captured step can always run warm
            N.check(N.lib.lipasr_mlp_project_per_layer(m._plan, N.ptr(m._params), self.rho, N.ptr(self.v_state), 0, 48, N.ptr(self.sigmas),
                                                       N.stream_ptr()))
            self._warm = True

    def profile_train(self, n_steps):
        """Time the classifier part (attack + fwd/bwd [+ all-reduce] + Adam + projection) of the next ``n_steps`` steps with
        HIP events recorded on the training stream around its launches; ``train_ms()`` returns the mean.  The MFCC of the
        next batch runs beside it on its own stream, so this is the classifier's time INSIDE the overlapped step."""
        self._prof = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(int(n_steps))]
        self._prof_i = 0

    def train_ms(self):
        used = self._prof[:self._prof_i]
        self._prof = None
        if not used:
            return 0.0
        used[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in used) / len(used)

    def step(self, waves, y_onehot, features=None, global_batch=None):
        """waves: float32 device tensor [b, n_samp] (a view into a resident pool is fine), y_onehot [b, classes].
        features: pre-extracted, already standardised [b, 20*L] features instead of waveforms (BASELINE config 2,
        the reference's own train_constraints.py flow); the MFCC stage is skipped.
        global_batch: data parallel with UNEVEN shards only (e.g. the last partial batch cut by ``shard_bounds``): the
        number of rows all ranks process in this step; default = this rank's rows x world size.
        Asynchronous: ``pipe.feats`` / ``pipe.labels`` / the model's buffers are valid after ``synchronize()``."""
        bsz = (features if features is not None else waves).shape[0]
        gb = int(global_batch) if global_batch is not None else bsz * self.dp.world
        b = self._i % self._nbuf
        self._i += 1
        self.feats, self.labels = self._feats2[b], self._labels2[b]
        if self._closed:
            raise RuntimeError("TrainPipeline.step() after close()")
        self._check_flags()
        # the inputs were produced on the caller's stream (an H2D copy, a noise kernel, a slice of a pool): order the
        # extraction stream after it.  Below, the caller's stream is in turn ordered after this step's extraction, so a
        # temporary handed to step() and freed right after it is not recycled (the caching allocator re-issues a block on
        # the stream it was allocated on) while the MFCC kernels still read it.  (Not record_stream(): the allocator would
        # then record events on the CU-masked stream whenever such a block is freed -- also after close() destroyed it.)
        caller = torch.cuda.current_stream(self.dev)
        xs = self.mfcc_stream
        idle = self.stream.query() if self._flags is not None else (self._last_train_ev is None or self._last_train_ev.query())
        wide = self._wide_stream is not None and features is None and idle
        if wide:  # nothing runs on the classifier's CUs: this extraction may have them
            xs = self._wide_stream
            self.ex.set(1, self._n_cu)
        if self.sync_inputs:
            xs.wait_stream(caller)
        with torch.cuda.stream(xs):
            if self._flags is not None:
                if self._last_xs is not None and self._last_xs is not xs:
                    xs.wait_stream(self._last_xs)  # the plan's scratch is shared: order this extraction after the last one on the other stream
                self._last_xs = xs
            elif self._ev_last_mfcc is not None and self._wide_stream is not None:
                xs.wait_event(self._ev_last_mfcc)
            if self._flags is not None:
                fl = self._flags
                if self._i > self._nbuf:
                    N.check(N.lib.lipasr_flag_wait(self.h.h, fl[self._nbuf + b:].data_ptr(), self._i - self._nbuf, self._flag_timeout_ms, self._flag_err.data_ptr(), N.stream_ptr()))
            elif self._ev_free[b] is not None:
                xs.wait_event(self._ev_free[b])  # the step that last read this buffer is done
            if features is not None:
                self._feats2[b][:bsz].copy_(features)
            else:
                if self._custom_ex:
                    self.ex(waves, self.mean, self.scale, out=self._feats2[b][:bsz])
                else:
                    self.ex(waves, self.L, self.mean, self.scale, out=self._feats2[b][:bsz])
            self._labels2[b][:bsz].copy_(y_onehot)
            if self._flags is not None:
                N.check(N.lib.lipasr_flag_signal(self.h.h, self._flags[b:].data_ptr(), self._i, N.stream_ptr()))
            else:
                self._ev_feat[b].record(xs)
                self._ev_last_mfcc = self._ev_feat[b]
        if wide:
            self.ex.set(1, self.mfcc_cus)
        if self.sync_inputs:
            if self._flags is not None:
                caller.wait_stream(xs)
            else:
                caller.wait_event(self._ev_feat[b])
        with torch.cuda.stream(self.stream):
            self._warm_start()
            if self._flags is not None:
                N.check(N.lib.lipasr_flag_wait(self.h.h, self._flags[b:].data_ptr(), self._i, self._flag_timeout_ms, self._flag_err.data_ptr(), N.stream_ptr()))
            else:
                self.stream.wait_event(self._ev_feat[b])
            prof = None
            if getattr(self, "_prof", None) is not None and self._prof_i < len(self._prof):
                prof = self._prof[self._prof_i]
                self._prof_i += 1
                prof[0].record(self.stream)
            if self.sync_bn:
                self._syncbn_step(bsz, b, gb)
            elif not self.use_graph:
                if self.dp.world == 1:
                    self._attack_and_train(bsz, b, gb)
                    self._update()
                elif self.overlap_buckets:
                    self._attack_and_train(bsz, b, gb, defer_dw0=True)
                    self._reduce_and_update_eager(bsz, b)
                else:
                    self._attack_and_train(bsz, b, gb)
                    self.dp.allreduce_grads(self.model._grads)
                    self._update()
            else:
                g = self._graphs.get((bsz, b, gb))
                if g is None:
                    if self.dp.world == 1:
                        g = (self._capture(lambda: (self._attack_and_train(bsz, b, gb), self._update())),)
                    elif self.overlap_buckets:
                        # three graphs around the two gradient buckets (reduce what is ready while the backward
                        # pass still computes): [attack + fwd/bwd without dW_0] | [dW_0] | [Adam + projection]
                        g = (self._capture(self._attack_and_train, bsz, b, gb, True), self._capture(self._dw0, bsz, b),
                             self._capture(self._update))
                    else:
                        g = (self._capture(self._attack_and_train, bsz, b, gb), self._capture(self._update))
                    self._graphs[(bsz, b, gb)] = g
                N.check(N.lib.lipasr_graph_launch(self.h.h, g[0], N.stream_ptr()))
                if len(g) == 3:
                    late = self._late
                    ha = self.dp.allreduce_async(self.model._grads[late:])   # 44 % of the bytes, beside the dW_0 GEMM
                    N.check(N.lib.lipasr_graph_launch(self.h.h, g[1], N.stream_ptr()))
                    hb = self.dp.allreduce_async(self.model._grads[:late])
                    ha.wait()
                    hb.wait()
                    N.check(N.lib.lipasr_graph_launch(self.h.h, g[2], N.stream_ptr()))
                elif len(g) == 2:
                    self.dp.allreduce_grads(self.model._grads)
                    N.check(N.lib.lipasr_graph_launch(self.h.h, g[1], N.stream_ptr()))
            if prof is not None:
                prof[1].record(self.stream)
            if self._flags is not None:
                N.check(N.lib.lipasr_flag_signal(self.h.h, self._flags[self._nbuf + b:].data_ptr(), self._i, N.stream_ptr()))
            else:
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self._ev_free[b] = ev
                self._last_train_ev = ev

    def synchronize(self):
        if getattr(self, "_wide_stream", None) is not None:
            self._wide_stream.synchronize()
        self.mfcc_stream.synchronize()
        self.stream.synchronize()
        self._check_flags()
        if self.model.exchange_errors():
            raise RuntimeError("TrainPipeline: a BatchNorm exchange inside a GEMM gave up after 2 s (workgroups of one column block were never "
                               "resident together: is the plan's CU budget larger than its stream's CU mask?); the step's results are invalid. "
                               "LIPASR_FUSE_BN=0 uses the launch chain")

    @property
    def mfcc_stream_kind(self):
        """"masked" (its own hardware queue, confined to ``mfcc_cus`` CUs) or "shared" (a pool stream on every CU)."""
        return "masked" if self._masked_stream is not None else "shared"

    def close(self):
        """Drains both streams, destroys this pipeline's graph executables and gives the CU-masked stream (a hardware
        queue of its own) back, in that order.  Idempotent; also reached from ``lipasr._native.shutdown`` at interpreter
        exit, so no HIP object of the pipeline is left to the runtime's static destructors."""
        if getattr(self, "_closed", True):
            return
        if N._capture_depth > 0:  # a finaliser in the middle of another pipeline's capture: synchronising / freeing would break it
            N._deferred.append((self.close, ()))
            return
        self._closed = True
        if not self.h.alive:  # the handle went first and took graphs and streams with it
            self._graphs.clear()
            self._masked_stream = self._masked_train_stream = None
            # the ExternalStream wrappers point at destroyed queues: nothing may synchronise on them any more
            self.stream = self.mfcc_stream = torch.cuda.current_stream(self.dev)
            return
        pending = None
        try:
            self.synchronize()
        except RuntimeError as e:  # a device-side wait reported: finish the teardown (queues, graphs), then tell the caller
            pending = e
        for g in self._graphs.values():
            for gid in g:
                N.lib.lipasr_graph_destroy(self.h.h, gid)
        self._graphs.clear()
        st = self._masked_stream
        if st is not None:
            self._masked_stream = None
            self.mfcc_stream = torch.cuda.Stream(device=self.dev)
            N.lib.lipasr_stream_destroy(self.h.h, st)
        st = self._masked_train_stream
        if st is not None:
            self._masked_train_stream = None
            self.stream = torch.cuda.Stream(device=self.dev)
            N.lib.lipasr_stream_destroy(self.h.h, st)
        if self._gemm_tiles_set and getattr(self.model, "_plan", None):
            self._gemm_tiles_set = False
            N.lib.lipasr_mlp_set_gemm_tiles(self.model._plan, 0)  # back to the library's default threshold
            N.lib.lipasr_mlp_set_cu_budget(self.model._plan, 0)
        if not self._custom_ex:
            self.ex.close()  # the pipeline's own MFCC plan
        if pending is not None:
            raise pending

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
