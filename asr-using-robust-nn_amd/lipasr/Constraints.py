"""The reference's Constraints.py surface (Constraints.py:9-189) over liblipasr's K3 kernels.

Same class names, constructor arguments, methods and layer protocol (``'dense' in layer.name``,
``get_weights()`` / ``set_weights()``) as the reference, so a driver written against it runs
unchanged:

    model.fit(..., callbacks=[simple_norm_constraint(rho=0.1, affected_layers_indices=[]), ...])

With a ``lipasr.keras.Model`` the projection runs in-stream on the device-resident kernels (no host
copy, no SVD).  With any other object that follows the layer protocol the kernels are staged to the
GPU, projected by the same HIP kernels and handed back through ``set_weights``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as N
from .keras import Callback, Model


def _dense_layers(model):
    return [l for l in model.layers if "dense" in l.name]  # Constraints.py:18,29,146,154


class _Staged:
    """Dense kernels of a model as device tensors: views for lipasr models, uploads otherwise."""

    def __init__(self, model):
        self.model = model
        self.native = isinstance(model, Model)
        self.layers = _dense_layers(model)
        if self.native:
            self.kernels = [l.kernel for l in self.layers]
            self.device = model._device
        else:
            self.device = torch.device("cuda", torch.cuda.current_device())
            self.kernels = [torch.as_tensor(np.asarray(l.get_weights()[0], dtype=np.float32)).to(self.device).contiguous() for l in self.layers]
        self.h = N.get_handle(self.device.index)
        self.rows = N.int_array([k.shape[0] for k in self.kernels])
        self.cols = N.int_array([k.shape[1] for k in self.kernels])
        self.ptrs = N.ptr_array([k.data_ptr() for k in self.kernels])

    def write_back(self, indices=None):
        if self.native:
            return
        for i, l in enumerate(self.layers):
            if indices is None or i in indices:
                b = l.get_weights()[1]
                l.set_weights([self.kernels[i].cpu().numpy(), b])


class norm_constraint(Callback):
    """Constraints.py:9-33: every Dense kernel <- max(w,0) * rho^(1/m) / (sigma_max + eps) after each batch.

    sigma_max comes from a warm-started power iteration (``iters`` round trips per batch after a
    ``cold_iters`` first call) instead of a full LAPACK SVD."""

    def __init__(self, rho, iters=4, cold_iters=48):
        super().__init__()
        self.rho = rho
        self.m = 0
        self.iters, self.cold_iters = int(iters), int(cold_iters)
        self._v = None
        self._warm = False
        self.last_sigmas = None

    def on_train_begin(self, logs=None):
        self.m = len(_dense_layers(self.model))  # Constraints.py:15-20
        self._v, self._warm = None, False

    def get_projection(self, w, rho):
        """Constraints.py:22-25 for one kernel given as an array; returns a NumPy array."""
        if self.m == 0:
            self.m = len(_dense_layers(self.model))
        dev = torch.device("cuda", torch.cuda.current_device())
        wt = torch.as_tensor(np.asarray(w, dtype=np.float32)).to(dev).contiguous().clone()
        v = torch.zeros(wt.shape[1], device=dev)
        sig = torch.zeros(1, device=dev)
        h = N.get_handle(dev.index)
        ptrs, rows, cols = N.ptr_array([wt.data_ptr()]), N.int_array([wt.shape[0]]), N.int_array([wt.shape[1]])
        # one layer, but the exponent is 1/m of the whole model: scale rho accordingly
        N.check(N.lib.lipasr_project_per_layer(h.h, C.cast(ptrs, N.PV), rows, cols, 1, float(rho) ** (1.0 / self.m), N.ptr(v), 0,
                                               self.cold_iters, N.ptr(sig), N.stream_ptr()))
        return wt.cpu().numpy()

    def on_batch_end(self, batch, logs=None):
        st = _Staged(self.model)
        if self.m == 0:
            self.m = len(st.layers)
        n_v = sum(k.shape[1] for k in st.kernels)
        if self._v is None or self._v.numel() != n_v or self._v.device != st.device:
            self._v = torch.zeros(n_v, device=st.device)
            self._sig = torch.zeros(len(st.kernels), device=st.device)
            self._warm = False
        iters = self.iters if self._warm else self.cold_iters
        N.check(N.lib.lipasr_project_per_layer(st.h.h, C.cast(st.ptrs, N.PV), st.rows, st.cols, len(st.kernels), float(self.rho),
                                               N.ptr(self._v), 1 if self._warm else 0, iters, N.ptr(self._sig), N.stream_ptr()))
        self._warm = True
        self.last_sigmas = self._sig
        st.write_back()


class customConstraint:
    """Constraints.py:38-49: w <- max(w,0) * rho / (||max(w,0)||_F + eps).  ``tf.norm(w, ord=2)`` with
    axis=None flattens, so the norm is Frobenius, and that is what is reproduced."""

    def __init__(self, rho):
        self.rho = rho

    def __call__(self, w):
        was_tensor = torch.is_tensor(w)
        dev = w.device if was_tensor and w.is_cuda else torch.device("cuda", torch.cuda.current_device())
        wt = (w if was_tensor else torch.as_tensor(np.asarray(w, dtype=np.float32))).to(device=dev, dtype=torch.float32).contiguous().clone()
        h = N.get_handle(dev.index)
        N.check(N.lib.lipasr_frobenius_project(h.h, N.ptr(wt), wt.numel(), float(self.rho), N.stream_ptr()))
        return wt if was_tensor else wt.cpu().numpy()

    def project_(self, w):
        """In-place variant on a device tensor (used when applied as a per-step kernel constraint)."""
        h = N.get_handle(w.device.index)
        N.check(N.lib.lipasr_frobenius_project(h.h, N.ptr(w), w.numel(), float(self.rho), N.stream_ptr()))
        return w

    def get_config(self):
        return {"rho": self.rho}


class simple_norm_constraint(Callback):
    """Constraints.py:135-189: rescale kernels so that ||W_m^T ... W_1^T||_2 moves towards rho.

    ``affected_layers_indices`` empty: every Dense layer in order, each projection seeing the
    already-rescaled earlier ones (:173-179); otherwise the listed indices, visited from the last
    Dense layer to the first, once per occurrence in the list (:181-189)."""

    def __init__(self, rho, affected_layers_indices):
        super().__init__()
        self.rho = rho
        self.m = 0
        self.affected_layers_indices = affected_layers_indices
        self.last_norms = None

    def get_w_list(self):
        return [l.get_weights()[0] for l in _dense_layers(self.model)]

    def get_layer_list(self):
        return _dense_layers(self.model)

    def _visit_order(self, n_layers):
        if len(self.affected_layers_indices) == 0:
            return list(range(n_layers))
        order = []
        for index in reversed(range(n_layers)):
            for layer_index in self.affected_layers_indices:
                if layer_index == index:
                    order.append(index)
        return order

    def _product_norm(self, st):
        sig = torch.zeros(1, device=st.device)
        N.check(N.lib.lipasr_product_norm(st.h.h, C.cast(st.ptrs, N.PV), st.rows, st.cols, len(st.kernels), N.ptr(sig), N.stream_ptr()))
        return sig

    def get_projection(self, w):
        """Constraints.py:158-169: w * (rho / (||cst||_2 + eps))^(1/len(w_list)) with the live kernels."""
        st = _Staged(self.model)
        n = float(self._product_norm(st).item())
        s = np.power(self.rho / (n + np.spacing(1)), 1.0 / len(st.kernels))
        return (np.asarray(w, dtype=np.float32) * np.float32(s)).astype(np.float32)

    def on_batch_end(self, batch, logs=None):
        st = _Staged(self.model)
        order = self._visit_order(len(st.kernels))
        norms = torch.zeros(len(order) + 1, device=st.device)
        N.check(N.lib.lipasr_project_product(st.h.h, C.cast(st.ptrs, N.PV), st.rows, st.cols, len(st.kernels), float(self.rho),
                                             N.int_array(order), len(order), N.ptr(norms), N.stream_ptr()))
        self.last_norms = norms
        st.write_back(set(order))


class norm_constraint_FISTA(Callback):
    """Constraints.py:54-130: dual forward-backward projection with singular-value clipping.

    Every step runs on the device: the products through lipasr_gemm_f32 (fp32 MFMA), the thin SVDs of the
    (classes x n_0) iterates and their clipping through lipasr_sv_clip (Gram + Jacobi in fp64), the step size
    from the spectral norms of A (lipasr_sv_clip) and B (lipasr_sigma_max).  Two scalars per iteration come
    back to the host for the reference's early-exit test (:91)."""

    def __init__(self, rho, nit):
        super().__init__()
        self.rho = rho
        self.m = 0
        self.nit = nit

    def get_w_list(self):
        return [l.get_weights()[0] for l in _dense_layers(self.model)]

    @staticmethod
    def _mm(a, b):
        """a @ b on the device through the fp32 MFMA GEMM."""
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty(a.shape[0], b.shape[1], device=a.device)
        h = N.get_handle(a.device.index)
        N.check(N.lib.lipasr_gemm_f32(h.h, 0, 0, a.shape[0], b.shape[1], a.shape[1], N.ptr(a), a.shape[1], N.ptr(b), b.shape[1],
                                      N.ptr(out), b.shape[1], N.stream_ptr()))
        return out

    @staticmethod
    def _sv_clip(x, hi=None, want_out=True):
        """(U min(S, hi) V^T, S) of a (R <= 32) x n device matrix; ``hi=None`` leaves S alone."""
        x = x.contiguous()
        h = N.get_handle(x.device.index)
        out = torch.empty_like(x) if want_out else None
        sv = torch.empty(x.shape[0], device=x.device)
        N.check(N.lib.lipasr_sv_clip(h.h, N.ptr(x), x.shape[0], x.shape[1], float("inf") if hi is None else float(hi),
                                     N.ptr(out) if want_out else None, N.ptr(sv), N.stream_ptr()))
        return out, sv

    @classmethod
    def _norm2(cls, a, iters=300):
        """np.linalg.norm(a, ord=2) (Constraints.py:72) on the device."""
        if a.shape[0] == a.shape[1] and getattr(a, "_lipasr_identity", False):
            return 1.0
        if min(a.shape) <= 32:
            x = a if a.shape[0] <= 32 else a.t()
            return float(cls._sv_clip(x, want_out=False)[1][0])
        a = a.contiguous()
        h = N.get_handle(a.device.index)
        v = torch.empty(a.shape[1], device=a.device)
        out = torch.empty(1, device=a.device)
        N.check(N.lib.lipasr_sigma_max(h.h, N.ptr(a), a.shape[0], a.shape[1], N.ptr(v), 0, iters, 0, N.ptr(out), N.stream_ptr()))
        return float(out)

    def Constraint_Fista(self, w, Y0, A, B, nit, rho):
        """Constraints.py:69-94 on device tensors; ``w`` is the transposed kernel (out, in)."""
        mm = self._mm
        Y, Yold = Y0, Y0
        gam = float(1.0 / ((self._norm2(A) * self._norm2(B) + np.spacing(1)) ** 2))
        alpha = 2.1
        w_new = w
        for i in range(nit):
            eta = i / (i + 1 + alpha)
            Z = Y + eta * (Y - Yold)
            Yold = Y
            w_new = w - mm(mm(A.t(), Z), B.t())
            w_new = w_new * (w_new >= 0)
            T = mm(mm(A, w_new), B)
            s = self._sv_clip(T, want_out=False)[1]  # :78-79 singular values of T
            criterion = float(torch.linalg.norm(w_new - w))
            constraint = float(torch.linalg.norm(torch.clamp(s - rho, min=0.0)))  # :81 ||s[s > rho] - rho||_2
            Yt = Z + gam * T
            # :86-89  Yt - gam * U clip(S / gam, 0, rho) V^T  ==  Yt - U min(S, rho * gam) V^T  with U S V^T = svd(Yt)
            Y = Yt - self._sv_clip(Yt, hi=rho * gam)[0]
            if criterion < 30 and constraint < 0.01:
                return w_new
        return w_new

    def get_projection(self, w, w_list=None, w_index=None):
        dev = torch.device("cuda", torch.cuda.current_device())
        if w_list is None:
            w_list = self.get_w_list()
        ws = [torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev) if not torch.is_tensor(x) else x for x in w_list]
        if w_index is None:
            wt = torch.as_tensor(np.asarray(w, dtype=np.float32)).to(dev) if not torch.is_tensor(w) else w
            for index in reversed(range(len(ws))):  # Constraints.py:100-102: located by value, lowest index wins
                if ws[index].shape == wt.shape and torch.equal(ws[index], wt):
                    w_index = index
            if w_index is None:
                raise ValueError("kernel not found in the model")
        wk = ws[w_index]
        A = None
        B = None
        for index in reversed(range(len(ws))):
            if index > w_index:
                A = ws[index].t().contiguous() if A is None else self._mm(A, ws[index].t())
            elif index < w_index:
                B = ws[index].t().contiguous() if B is None else self._mm(B, ws[index].t())
        if w_index == 0:
            B = torch.eye(wk.shape[0], device=dev)
            B._lipasr_identity = True
        if w_index == len(ws) - 1:
            A = torch.eye(wk.shape[1], device=dev)
            A._lipasr_identity = True
        Y0 = torch.zeros(A.shape[0], B.shape[1], device=dev)
        return self.Constraint_Fista(wk.t().contiguous(), Y0, A, B, self.nit, self.rho)

    def on_batch_end(self, batch, logs=None):
        st = _Staged(self.model)
        for i in range(len(st.kernels)):
            w_new = self.get_projection(None, w_list=st.kernels, w_index=i)
            st.kernels[i].copy_(w_new.t())
        st.write_back()
