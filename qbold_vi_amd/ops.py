"""Torch-tensor front-end of the C ABI: device buffers in, device buffers out.

PyTorch owns the memory and the stream; every number is produced by libqbold_hip.so.  Inputs must
be float32 CUDA (HIP) tensors -- CPU tensors are rejected, there is no host fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Consts, EncoderShape, Geometry, LossCfg, QboldError

PARAM_KEYS = ("gamma", "b0", "dchi", "te", "r2t", "tr", "ti", "t1b", "hct",
              "tau_start", "tau_end", "tau_step")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name, shape_last=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise QboldError(f"{name}: tensor is on {t.device}; the HIP kernels need a cuda (ROCm) tensor "
                         "and there is no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if shape_last is not None and t.shape[-1] != shape_last:
        raise ValueError(f"{name}: last dimension is {t.shape[-1]}, expected {shape_last}")
    return t


def merge_ranges(pieces):
    """Sorted [offset, length] pieces with touching neighbours merged."""
    out = []
    for off, n in sorted(pieces):
        if out and out[-1][0] + out[-1][1] == off:
            out[-1][1] += n
        else:
            out.append([off, n])
    return [tuple(r) for r in out]


class EncoderWeights:
    """The canonical (Keras-orientation) weight blob of the voxel-wise encoder plus its
    MFMA-ordered device copy.  Layout: include/qbold_hip.h, qbold_encoder_num_params."""

    NAMES = ("W0", "b0", "Wc", "bc", "Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Wf", "bf", "Ws", "bs")

    PRECISIONS = {"f32": 0, "bf16": 1}   # qbold_encoder_precision
    ACTIVATIONS = {"relu": 0, "gelu": 1}  # qbold_activation

    def __init__(self, ctx, T, U, L, channelwise_gating=True, gate_offset=0.0, spatial_taps=1,
                 precision="f32", activation="relu", layer_norm=False, dropout_rate=0.0):
        self.ctx = ctx
        if activation not in self.ACTIVATIONS:
            raise ValueError(f"activation {activation!r}: the kernels implement 'relu' and 'gelu'")
        if not 0.0 <= float(dropout_rate) < 1.0:
            raise ValueError("dropout_rate must lie in [0, 1)")
        # layer_norm / dropout_rate: EncoderTrainer's use_layer_norm / dropout_rate (model.py:131-140); dropout_seed is
        # set per training step by TrainState and stays 0 (inference: identity) everywhere else
        self.shape = EncoderShape(int(T), int(U), int(L), int(bool(channelwise_gating)),
                                  float(gate_offset), 9 if spatial_taps == 9 else 1,
                                  self.PRECISIONS[precision], self.ACTIVATIONS[activation],
                                  int(bool(layer_norm)), float(dropout_rate), 0)
        if precision != "f32" and not Context.fits_fused(self.shape):
            raise ValueError("precision='bf16' exists for the fused voxel kernels (U <= 64, L <= 2, T in {11, 24})")
        lib = _lib.load()
        self.num_params = int(lib.qbold_encoder_num_params(C.byref(self.shape)))
        self.flat = torch.zeros(self.num_params, dtype=torch.float32, device=ctx.device)
        self.packed = torch.zeros(int(lib.qbold_encoder_packed_floats(C.byref(self.shape))),
                                  dtype=torch.float32, device=ctx.device)
        self._dirty = {"packed", "wide", "fused"}   # device images to rebuild from the canonical blob
        # weight-streaming image for widths beyond the LDS-resident kernels (U = 128 / 256)
        n_wide = int(lib.qbold_encoder_wide_packed_floats(C.byref(self.shape)))
        # gelu runs on the general layer-wise kernels only (include/qbold_hip.h, qbold_activation)
        self.wide = (n_wide > 0 and not Context.fits_fused(self.shape) and self.shape.activation == 0
                     and not self.shape.layer_norm)
        self.wide_packed = (torch.zeros(n_wide, dtype=torch.float32, device=ctx.device)
                            if self.wide else None)
        self._wide_ws = None
        self._wide_ws_n = 0
        # one-launch form of the wide stream-2 encoder (wide_fused_kernels.hip: activations in registers)
        n_fused = int(lib.qbold_encoder_fused_packed_floats(C.byref(self.shape)))
        self.fused_wide = self.wide and n_fused > 0
        self.fused_packed = (torch.zeros(n_fused, dtype=torch.float32, device=ctx.device)
                             if self.fused_wide else None)

    # canonical blob views -------------------------------------------------------------------
    def _slices(self):
        T, U, L = self.shape.T, self.shape.U, self.shape.L
        G = U if self.shape.channelwise_gating else 1
        rs = (3, 3, U, U) if self.shape.spatial_taps == 9 else (U, U)  # residual conv kernels
        out, off = {}, 0

        def take(name, *shape):
            nonlocal off
            n = int(np.prod(shape))
            out.setdefault(name, []).append((off, shape))
            off += n
        take("W0", T, U); take("b0", U)
        for _ in range(L):
            take("Wc", U, U); take("bc", U); take("Wr1", *rs); take("br1", U)
            take("Wr2", *rs); take("br2", U); take("Wg", U, G); take("bg", G)
        take("Wf", U, 5); take("bf", 5); take("Ws", U, T); take("bs", T)
        if self.shape.layer_norm:   # GroupNormalization gamma1, beta1, gamma2, beta2 per block, behind the heads
            for _ in range(L):
                take("ln", 4, U)
        assert off == self.num_params
        return out

    def set_from_arrays(self, w):
        """w: dict name -> array; per-block tensors carry a leading [L] axis."""
        sl = self._slices()
        flat = torch.empty(self.num_params, dtype=torch.float32)
        for name, pieces in sl.items():
            if name == "ln" and name not in w:   # gamma = 1, beta = 0: GroupNormalization's initialisers
                w = dict(w, ln=np.tile(np.array([1.0, 0.0, 1.0, 0.0], np.float32)[None, :, None],
                                       (self.shape.L, 1, self.shape.U)))
            arr = torch.as_tensor(np.asarray(w[name], dtype=np.float32))
            for l, (off, shape) in enumerate(pieces):
                src = arr[l] if len(pieces) > 1 or arr.dim() == len(shape) + 1 else arr
                flat[off:off + int(np.prod(shape))] = src.reshape(-1)
        self.flat.copy_(flat.to(self.flat.device))
        self.mark_dirty()
        return self

    def to_arrays(self):
        sl = self._slices()
        flat = self.flat.detach().cpu().numpy()
        out = {}
        for name, pieces in sl.items():
            arrs = [flat[off:off + int(np.prod(shape))].reshape(shape) for off, shape in pieces]
            out[name] = np.stack(arrs) if name in ("Wc", "bc", "Wr1", "br1", "Wr2", "br2", "Wg", "bg", "ln") \
                else arrs[0]
        return out

    def mark_dirty(self):
        self._dirty = {"packed", "wide", "fused"}

    # the tensors stream 1 (the pre-training output, model.py:199) depends on; every other variable gets a None
    # gradient from Keras' loss=[synthetic_data_loss, None, None] (train.py:388-392) and is left alone by
    # apply_gradients -- no Adam update AND no decoupled weight decay
    STREAM1_TENSORS = ("W0", "b0", "Wc", "bc", "Wf", "bf")

    def param_ranges(self, names):
        """Merged, sorted [offset, length] ranges of the named tensors inside the canonical blob."""
        return merge_ranges([(off, int(np.prod(shape))) for n in names for off, shape in self._slices()[n]])

    def set_precision(self, precision):
        """'f32' (split-f16 MFMA, float32-grade) or 'bf16' (single bf16 MFMA pass) for the fused
        voxel kernels; the MFMA-ordered image is rebuilt on next use."""
        if precision != "f32" and not Context.fits_fused(self.shape):
            raise ValueError("precision='bf16' exists for the fused voxel kernels (U <= 64, L <= 2, T in {11, 24})")
        self.shape.precision = self.PRECISIONS[precision]
        self.mark_dirty()
        return self

    def wide_ptr(self):
        if "wide" in self._dirty:
            _lib.check(_lib.load().qbold_encoder_wide_pack(self.ctx.handle, C.byref(self.shape), _ptr(self.flat),
                                                           _ptr(self.wide_packed), _stream()),
                       "qbold_encoder_wide_pack")
            self._dirty.discard("wide")
        return _ptr(self.wide_packed)

    def fused_ptr(self):
        if "fused" in self._dirty:
            _lib.check(_lib.load().qbold_encoder_fused_pack(self.ctx.handle, C.byref(self.shape), _ptr(self.flat),
                                                            _ptr(self.fused_packed), _stream()),
                       "qbold_encoder_fused_pack")
            self._dirty.discard("fused")
        return _ptr(self.fused_packed)

    def wide_workspace(self, N):
        if self._wide_ws is None or self._wide_ws_n < N:
            n = int(_lib.load().qbold_encoder_wide_workspace_floats(C.byref(self.shape), int(N)))
            self._wide_ws = torch.empty(n, dtype=torch.float32, device=self.flat.device)
            self._wide_ws_n = N
        return self._wide_ws

    def packed_ptr(self):
        if "packed" in self._dirty:
            _lib.check(_lib.load().qbold_encoder_pack(self.ctx.handle, C.byref(self.shape),
                                                      _ptr(self.flat), _ptr(self.packed), _stream()),
                       "qbold_encoder_pack")
            self._dirty.discard("packed")
        return _ptr(self.packed)


class Context:
    """One qbold_ctx: folded constants + F(x) table on one device."""

    def __init__(self, params, full_model=True, include_blood=True, multi_image_normalisation=False,
                 predict_log_data=False, student_t_df=None, device=None, host_only=False):
        lib = _lib.load()
        consts = Consts(**{k: float(params[k]) for k in PARAM_KEYS},
                        full_model=int(bool(full_model)), include_blood=int(bool(include_blood)))
        use_t = student_t_df is not None and student_t_df < 50  # model.py:557
        loss = LossCfg(int(bool(multi_image_normalisation)), int(bool(predict_log_data)), int(use_t),
                       float(student_t_df) if use_t else 0.0)
        if host_only:
            dev_index = -1
            self.device = torch.device("cpu")
        else:
            if not torch.cuda.is_available():
                raise QboldError("no ROCm device visible to PyTorch; libqbold_hip.so needs an "
                                 "MI355X (there is no CPU fallback)")
            self.device = torch.device(device if device is not None else
                                       f"cuda:{torch.cuda.current_device()}")
            dev_index = self.device.index if self.device.index is not None else 0
        h = C.c_void_p()
        _lib.check(lib.qbold_ctx_create(C.byref(consts), C.byref(loss), dev_index, C.byref(h)),
                   "qbold_ctx_create")
        self.handle = h
        self.lib = lib
        self.T = lib.qbold_ctx_num_taus(h)
        self.se_idx = lib.qbold_ctx_se_idx(h)
        taus = np.zeros(self.T, np.float32)
        _lib.check(lib.qbold_ctx_taus(h, taus.ctypes.data_as(C.c_void_p)), "qbold_ctx_taus")
        self.taus = taus
        self._ws = None
        self._sums = None
        self.force_layerwise_wide = False   # tests: compare the one-launch wide encoder with the layer-wise one

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            self.lib.qbold_ctx_destroy(h)
            self.handle = None

    # -- configuration ---------------------------------------------------------------------
    def set_tissue_mode(self, mode):
        mode = {"table": 0, "literal": 1}.get(mode, mode)
        _lib.check(self.lib.qbold_ctx_set_tissue_mode(self.handle, int(mode)), "set_tissue_mode")

    def set_kernel_selection(self, mask):
        """qbold_ctx_set_kernel_selection: dispatch to older EQUIVALENT kernels (QBOLD_KSEL_* of include/qbold_hip.h);
        0 restores the default, fastest forms.  Used by the tests that hold fused kernels to the layer-wise ones."""
        _lib.check(self.lib.qbold_ctx_set_kernel_selection(self.handle, int(mask)), "set_kernel_selection")

    def set_grad_node0(self, on):
        _lib.check(self.lib.qbold_ctx_set_grad_node0(self.handle, int(bool(on))), "set_grad_node0")

    def table_eval(self, x):
        x = np.ascontiguousarray(x, np.float32)
        F = np.empty_like(x)
        dF = np.empty_like(x)
        _lib.check(self.lib.qbold_ctx_table_eval(self.handle, x.ctypes.data_as(C.c_void_p),
                                                 F.ctypes.data_as(C.c_void_p),
                                                 dF.ctypes.data_as(C.c_void_p), x.size), "table_eval")
        return F, dF

    def _workspace(self):
        if self._ws is None:
            n = int(self.lib.qbold_elbo_workspace_bytes(self.handle))
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._ws

    # -- forward model ---------------------------------------------------------------------
    def signal_fwd(self, oef_dbv):
        x = _f32(oef_dbv, "oef_dbv", 2)
        V = x.numel() // 2
        out = torch.empty(x.shape[:-1] + (self.T,), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.qbold_signal_fwd(self.handle, _ptr(x), _ptr(out), V, _stream()),
                   "qbold_signal_fwd")
        return out

    def signal_fwd_ex(self, oef_dbv, hct=None, alt=None, from_index=None):
        """signals.py:64-96 options: per-voxel hct [V], misalignment (alt [V,2], from_index [V] int32)."""
        x = _f32(oef_dbv, "oef_dbv", 2)
        V = x.numel() // 2
        h = a = f = None
        if hct is not None:
            h = _f32(hct, "hct")
            if h.numel() != V:
                raise ValueError("hct must hold one value per (OEF, DBV) pair")
        if (alt is None) != (from_index is None):
            raise ValueError("alt and from_index come together")
        if alt is not None:
            a = _f32(alt, "alt", 2)
            f = from_index.to(device=x.device, dtype=torch.int32).contiguous()
            if a.numel() != 2 * V or f.numel() != V:
                raise ValueError("alt [V,2] and from_index [V] must match oef_dbv")
        out = torch.empty(x.shape[:-1] + (self.T,), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.qbold_signal_fwd_ex(self.handle, _ptr(x), _ptr(h), _ptr(a), _ptr(f), _ptr(out), V,
                                                _stream()), "qbold_signal_fwd_ex")
        return out

    def signal_bwd(self, oef_dbv, grad_signal):
        x = _f32(oef_dbv, "oef_dbv", 2)
        g = _f32(grad_signal, "grad_signal", self.T)
        V = x.numel() // 2
        if g.numel() != V * self.T:
            raise ValueError("grad_signal does not match oef_dbv")
        out = torch.empty_like(x)
        _lib.check(self.lib.qbold_signal_bwd(self.handle, _ptr(x), _ptr(g), _ptr(out), V, _stream()),
                   "qbold_signal_bwd")
        return out

    # -- encoder ---------------------------------------------------------------------------
    @staticmethod
    def fits_fused(shape):
        """The LDS-resident fused kernels cover U <= 64, L <= 2, T in {11, 24}; anything else (e.g.
        BASELINE config 3: U = 256, T = 64) takes the layer-wise GEMM path."""
        return (shape.U <= 64 and shape.L <= 2 and shape.T in (11, 24) and shape.activation == 0   # gelu: layer-wise
                and not shape.layer_norm)                                                        # and so is GroupNorm

    def encoder_fwd(self, weights, x, want=("out1", "out2", "sigma")):
        x = _f32(x, "x", self.T)
        N = x.numel() // self.T
        lead = x.shape[:-1]
        if weights.wide:  # weight-streaming MFMA GEMM per layer (wide_kernels.hip)
            x2 = x.reshape(N, self.T)
            o1 = o2 = sg = None
            for sel, wanted in ((1, "out1" in want), (2, "out2" in want or "sigma" in want)):
                if not wanted:
                    continue
                q = torch.empty((N, 5), dtype=torch.float32, device=x.device)
                ls = torch.empty((N, self.T), dtype=torch.float32, device=x.device) if sel == 2 else None
                if sel == 2 and weights.fused_wide and not self.force_layerwise_wide:
                    _lib.check(self.lib.qbold_encoder_fused_fwd(self.handle, C.byref(weights.shape),
                                                                weights.fused_ptr(), _ptr(x2), _ptr(q), _ptr(ls), N,
                                                                _stream()), "qbold_encoder_fused_fwd")
                else:
                    _lib.check(self.lib.qbold_encoder_wide_fwd(self.handle, C.byref(weights.shape),
                                                               weights.wide_ptr(), _ptr(x2), sel,
                                                               _ptr(weights.wide_workspace(N)), _ptr(q), _ptr(ls), N,
                                                               _stream()), "qbold_encoder_wide_fwd")
                if sel == 1:
                    o1 = q.reshape(lead + (5,))
                else:
                    o2 = q.reshape(lead + (5,)) if "out2" in want else None
                    sg = self.transform("exp", ls).reshape(lead + (self.T,)) if "sigma" in want else None
            return o1, o2, sg
        if not self.fits_fused(weights.shape):
            st = getattr(weights, "_layerwise_state", None)
            if st is None:  # activation workspace is kept with the weights, not re-allocated per call
                st = weights._layerwise_state = TrainState(self, weights, optimiser_state=False)
                st.fused_forward = False
            x2 = x.reshape(N, self.T)
            o1 = o2 = sg = None
            if "out1" in want:
                o1 = st.forward(x2, 1)[0].reshape(lead + (5,))
            if "out2" in want or "sigma" in want:
                q2, ls = st.forward(x2, 2)
                o2 = q2.reshape(lead + (5,)) if "out2" in want else None
                sg = self.transform("exp", ls).reshape(lead + (self.T,)) if "sigma" in want else None
            return o1, o2, sg
        mk = lambda c: torch.empty(lead + (c,), dtype=torch.float32, device=x.device)
        o1 = mk(5) if "out1" in want else None
        o2 = mk(5) if "out2" in want else None
        sg = mk(self.T) if "sigma" in want else None
        _lib.check(self.lib.qbold_encoder_fwd(self.handle, C.byref(weights.shape), weights.packed_ptr(),
                                              _ptr(x), _ptr(o1), _ptr(o2), _ptr(sg), N, _stream()),
                   "qbold_encoder_fwd")
        return o1, o2, sg

    # -- logit-normal ----------------------------------------------------------------------
    def reparam(self, q, z):
        q = _f32(q, "q", 5)
        z = _f32(z, "z", 2)
        N = q.numel() // 5
        out = torch.empty(q.shape[:-1] + (2,), dtype=torch.float32, device=q.device)
        _lib.check(self.lib.qbold_reparam(self.handle, _ptr(q), _ptr(z), _ptr(out), N, _stream()),
                   "qbold_reparam")
        return out

    def logit_mvn_nlogp(self, y, params):
        y = _f32(y, "y", 2)
        p = _f32(params, "params", 5)
        N = p.numel() // 5
        out = torch.empty(p.shape[:-1], dtype=torch.float32, device=p.device)
        _lib.check(self.lib.qbold_logit_mvn_nlogp(self.handle, _ptr(y), _ptr(p), _ptr(out), N,
                                                  _stream()), "qbold_logit_mvn_nlogp")
        return out

    def posterior_moments(self, q, n_samples=20, z=None, seed=1, voxel0=0, want_vars=True):
        q = _f32(q, "q", 5)
        N = q.numel() // 5
        if z is not None:
            z = _f32(z, "z", 2)
        means = torch.empty(q.shape[:-1] + (3,), dtype=torch.float32, device=q.device)
        var = torch.empty_like(means) if want_vars else None
        _lib.check(self.lib.qbold_posterior_moments(self.handle, _ptr(q), _ptr(z), int(n_samples),
                                                    int(seed), int(voxel0), _ptr(means), _ptr(var),
                                                    N, _stream()), "qbold_posterior_moments")
        return means, var

    def r2p_loss_bwd(self, y_true, q, n_samples=10, z=None, seed=1, voxel0=0, scale=1.0, g_q=None,
                     loss_v=None, want_grad=True):
        """The R2' term of synthetic_data_loss (use_r2p_loss, model.py:475-490): ADDS the per-voxel value
        to loss_v [N] and scale * d / d q to g_q [N, 5] (fresh zero tensors when not given).  y_true
        [N, >= 3] carries the true R2' in column 2."""
        y = _f32(y_true, "y_true")
        q = _f32(q, "q", 5)
        N = q.numel() // 5
        if y.dim() != 2 or y.shape[0] != N or y.shape[1] < 3:
            raise ValueError("y_true must be [N, >= 3] (OEF, DBV, R2')")
        if z is not None:
            z = _f32(z, "z", 2)
            if z.numel() != N * n_samples * 2:
                raise ValueError("z must be [N, n_samples, 2]")
        if loss_v is None:
            loss_v = torch.zeros(N, dtype=torch.float32, device=q.device)
        if g_q is None and want_grad:
            g_q = torch.zeros((N, 5), dtype=torch.float32, device=q.device)
        _lib.check(self.lib.qbold_r2p_loss_bwd(self.handle, _ptr(y), int(y.shape[1]), _ptr(q), _ptr(z),
                                               int(n_samples), int(seed), int(voxel0), float(scale),
                                               _ptr(g_q), _ptr(loss_v), N, _stream()), "qbold_r2p_loss_bwd")
        return loss_v, g_q

    # -- ELBO ------------------------------------------------------------------------------
    def elbo_fwd(self, x, mask, q, prior, sigma, S=1, K=70, zs=None, zk=None, seed=1, voxel0=0,
                 per_voxel=True):
        """Returns (sums double[3] device tensor = (sum m*nll, sum [m>0] kl, sum m), nll_kl [N,2])."""
        x = _f32(x, "x", self.T)
        N = x.numel() // self.T
        q = _f32(q, "q", 5)
        prior = _f32(prior, "prior", 5)
        sigma = _f32(sigma, "sigma", self.T)
        mask = _f32(mask, "mask") if mask is not None else None
        zs = _f32(zs, "zs", 2) if zs is not None else None
        zk = _f32(zk, "zk", 2) if zk is not None else None
        if zs is not None and zs.numel() != N * S * 2:
            raise ValueError("zs must be [N, S, 2]")
        if zk is not None and zk.numel() != N * K * 2:
            raise ValueError("zk must be [N, K, 2]")
        sums = torch.empty(3, dtype=torch.float64, device=x.device)
        out = torch.empty((N, 2), dtype=torch.float32, device=x.device) if per_voxel else None
        _lib.check(self.lib.qbold_elbo_fwd(self.handle, _ptr(x), _ptr(mask), _ptr(q), _ptr(prior),
                                           _ptr(sigma), _ptr(zs), _ptr(zk), int(S), int(K), int(seed),
                                           int(voxel0), _ptr(out), _ptr(sums),
                                           _ptr(self._workspace()), N, _stream()), "qbold_elbo_fwd")
        return sums, out

    def vi_fwd_exact(self, weights, x, mask, prior, S=1, K=70, seed=1, voxel0=0):
        """The same evaluation with the encoder on the exact-float32 layer-wise path (f32-input MFMA GEMMs,
        qbold_encoder_train_fwd) -- no f16 operand split, so no 65504 operand limit.  Slow (one launch per layer,
        activations through HBM); the fallback of vi_fwd(range_check=True)."""
        x = _f32(x, "x", self.T)
        N = x.numel() // self.T
        st = getattr(weights, "_layerwise_state", None)
        if st is None:
            st = weights._layerwise_state = TrainState(self, weights, optimiser_state=False)
            st.fused_forward = False   # the exact-f32 GEMM forward: the whole point of this path
        q2, ls = st.forward(x.reshape(N, self.T), 2)
        sums, nk = self.elbo_fwd(x, mask, q2, prior, self.transform("exp", ls), S, K, seed=seed, voxel0=voxel0)
        return sums, q2, nk

    def vi_fwd(self, weights, x, mask, prior, S=1, K=70, seed=1, voxel0=0, want_q=True,
               per_voxel=True, out=None, range_check=False):
        """Fused encoder + ELBO.  Returns (sums, q [N,5] or None, nll_kl [N,2] or None).

        range_check: the MFMA encoders split every operand into two f16 halves (float32-grade for
        6.1e-5 <= |x| <= 65504, include/qbold_hip.h); an activation beyond 65504 overflows to inf and surfaces as
        NON-FINITE SUMS -- the kernels' status channel.  With range_check=True the sums are read back (one
        device synchronisation) and a non-finite result is recomputed on the exact-float32 layer-wise path,
        which reproduces the float32 reference wherever that is finite.  Off by default: the benchmark and the
        training steps do not pay the synchronisation; FineTuner.elbo (the reference-shaped API) turns it on."""
        x = _f32(x, "x", self.T)
        N = x.numel() // self.T
        prior = _f32(prior, "prior", 5)
        mask = _f32(mask, "mask") if mask is not None else None
        if range_check:
            sums, qo, nk = self.vi_fwd(weights, x, mask, prior, S, K, seed, voxel0, want_q, per_voxel, out)
            if N > 0 and not bool(torch.isfinite(sums).all()):
                sums2, q2, nk2 = self.vi_fwd_exact(weights, x, mask, prior, S, K, seed, voxel0)
                if bool(torch.isfinite(sums2).all()):   # the split overflowed, float32 did not
                    self.range_fallbacks = getattr(self, "range_fallbacks", 0) + 1
                    if out is not None:
                        out[0].copy_(sums2)
                        if out[1] is not None:
                            out[1].copy_(q2)
                        if out[2] is not None:
                            out[2].copy_(nk2)
                        return out[0], out[1], out[2]
                    return sums2, (q2 if want_q else None), (nk2 if per_voxel else None)
            return sums, qo, nk
        if weights.fused_wide and not self.force_layerwise_wide:
            nws = int(self.lib.qbold_vi_workspace_bytes(self.handle, C.byref(weights.shape), N))
            if nws > int(self.lib.qbold_elbo_workspace_bytes(self.handle)) + 256:  # the two-launch wide path applies
                ws = getattr(weights, "_vi_ws", None)
                if ws is None or ws.numel() < nws:
                    ws = weights._vi_ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
                if out is None:
                    sums = torch.empty(3, dtype=torch.float64, device=x.device)
                    qo = torch.empty((N, 5), dtype=torch.float32, device=x.device) if want_q else None
                    nk = torch.empty((N, 2), dtype=torch.float32, device=x.device) if per_voxel else None
                else:
                    sums, qo, nk = out
                _lib.check(self.lib.qbold_vi_fwd(self.handle, C.byref(weights.shape), weights.fused_ptr(),
                                                 _ptr(x), _ptr(mask), _ptr(prior), int(S), int(K), int(seed),
                                                 int(voxel0), _ptr(qo), _ptr(nk), _ptr(sums), _ptr(ws), N, _stream()),
                           "qbold_vi_fwd")
                return sums, qo, nk
        if not self.fits_fused(weights.shape):  # unfused composition of the same pieces
            _, q2, sg = self.encoder_fwd(weights, x.reshape(N, self.T), want=("out2", "sigma"))
            sums, nk = self.elbo_fwd(x, mask, q2, prior, sg, S, K, seed=seed, voxel0=voxel0)
            return sums, (q2 if want_q else None), (nk if per_voxel else None)
        if out is None:
            sums = torch.empty(3, dtype=torch.float64, device=x.device)
            qo = torch.empty((N, 5), dtype=torch.float32, device=x.device) if want_q else None
            nk = torch.empty((N, 2), dtype=torch.float32, device=x.device) if per_voxel else None
        else:
            sums, qo, nk = out
        _lib.check(self.lib.qbold_vi_fwd(self.handle, C.byref(weights.shape), weights.packed_ptr(),
                                         _ptr(x), _ptr(mask), _ptr(prior), int(S), int(K), int(seed),
                                         int(voxel0), _ptr(qo), _ptr(nk), _ptr(sums),
                                         _ptr(self._workspace()), N, _stream()), "qbold_vi_fwd")
        return sums, qo, nk


# --------------------------------------------------------------------------------------------
# the small pieces of the API surface (misc_kernels.hip)
# --------------------------------------------------------------------------------------------
TRANSFORM_OPS = {"exp": 6, "transform_std": 0, "transform_offdiag": 1, "inv_transform_std": 2,
                 "forward_transform": 3, "backwards_transform": 4, "backwards_transform_logit": 5}


def _ctx_method(fn):
    setattr(Context, fn.__name__, fn)
    return fn


@_ctx_method
def normalise(self, x):
    x = _f32(x, "x", self.T)
    out = torch.empty_like(x)
    _lib.check(self.lib.qbold_normalise(self.handle, _ptr(x), _ptr(out), x.numel() // self.T,
                                        _stream()), "qbold_normalise")
    return out


@_ctx_method
def transform(self, op, x):
    x = _f32(x, "x")
    out = torch.empty_like(x)
    _lib.check(self.lib.qbold_transform(self.handle, TRANSFORM_OPS[op], _ptr(x), _ptr(out), x.numel(),
                                        _stream()), "qbold_transform")
    return out


@_ctx_method
def synth_loss(self, y_true, q, inv_gamma_alpha=0.0, inv_gamma_beta=0.0):
    """Per-voxel pre-training loss (model.py:449-514) through qbold_synth_loss_bwd; the gradient
    output is discarded."""
    y = _f32(y_true, "y_true")
    q = _f32(q, "q", 5)
    N = q.shape[0]
    gq = torch.empty((N, 5), dtype=torch.float32, device=q.device)
    lv = torch.empty(N, dtype=torch.float32, device=q.device)
    _lib.check(self.lib.qbold_synth_loss_bwd(self.handle, _ptr(y), int(y.shape[-1]), _ptr(q), _ptr(gq), _ptr(lv),
                                             1.0, float(inv_gamma_alpha), float(inv_gamma_beta), N, _stream()),
               "qbold_synth_loss_bwd")
    return lv


@_ctx_method
def hyper_prior_bwd(self, q, ig, scale=1.0, g_q=None, loss_v=None, want_stats=True):
    """The learned inverse-gamma hyper-prior term (infer_inv_gamma, model.py:493-507) through
    qbold_hyper_prior_bwd: ig = (alpha_oef, beta_oef, alpha_dbv, beta_dbv) host floats.  ADDS the per-voxel term
    to loss_v and scale * d / d q to g_q; returns the device double[4] sums (sum log v_o, sum 1/v_o, sum log v_d,
    sum 1/v_d) or None."""
    q = _f32(q, "q", 5)
    N = q.shape[0]
    ig4 = (C.c_double * 4)(*[float(v) for v in ig])
    stats = torch.zeros(4, dtype=torch.float64, device=q.device) if want_stats else None
    _lib.check(self.lib.qbold_hyper_prior_bwd(self.handle, _ptr(q), ig4, float(scale), _ptr(g_q), _ptr(loss_v),
                                              _ptr(stats), N, _stream()), "qbold_hyper_prior_bwd")
    return stats


@_ctx_method
def kl_diag(self, q, prior, mask=None, g_q=None, per_voxel=True):
    """Closed-form KL of the diagonal family (use_mvg=False): q, prior [N,5] (columns 0-3 used).
    Returns (sums double[3] = (0, sum [m>0] kl, sum m), kl_v [N] or None); d kl / d q is added to g_q."""
    q = _f32(q, "q", 5)
    prior = _f32(prior, "prior", 5)
    N = q.numel() // 5
    mask = _f32(mask, "mask") if mask is not None else None
    sums = torch.empty(3, dtype=torch.float64, device=q.device)
    kl = torch.empty(N, dtype=torch.float32, device=q.device) if per_voxel else None
    _lib.check(self.lib.qbold_kl_diag(self.handle, _ptr(q), _ptr(prior), _ptr(mask), _ptr(kl), _ptr(g_q), _ptr(sums),
                                      _ptr(self._workspace()), N, _stream()), "qbold_kl_diag")
    return sums, kl


@_ctx_method
def kl_mog(self, q, comps, z=None, seed=1, voxel0=0):
    """kl_loss against a mixture-of-Gaussians population prior (model.py:666-685): q [N,5] (columns 0-3 used), comps
    [M,4] raw parameters; one draw per dimension from z [N,2] or the Philox stream.  Returns kl [N] (unmasked)."""
    q = _f32(q, "q", 5)
    comps = _f32(comps, "comps", 4)
    N = q.numel() // 5
    z = _f32(z, "z", 2) if z is not None else None
    if z is not None and z.numel() != 2 * N:
        raise ValueError("z must be [N, 2]")
    out = torch.empty(N, dtype=torch.float32, device=q.device)
    _lib.check(self.lib.qbold_kl_mog(self.handle, _ptr(q), _ptr(comps), comps.numel() // 4, _ptr(z), int(seed), int(voxel0),
                                     _ptr(out), N, _stream()), "qbold_kl_mog")
    return out


@_ctx_method
def wls_fit(self, signals, tau_min=0.016):
    """loglinear.fit_wls per voxel: signals [N,T] -> [N,3] = (OEF, DBV, R2'), clipped."""
    x = _f32(signals, "signals", self.T)
    out = torch.empty((x.shape[0], 3), dtype=torch.float32, device=x.device)
    _lib.check(self.lib.qbold_wls_fit(self.handle, _ptr(x), float(tau_min), _ptr(out), x.shape[0], _stream()),
               "qbold_wls_fit")
    return out


@_ctx_method
def nll_fwd(self, x, mask, pred, sigma, S=1):
    """Per-row NLL (before masking): x [N,T], pred/sigma [N*S,T] -> [N*S]."""
    x = _f32(x, "x", self.T)
    pred = _f32(pred, "pred", self.T)
    sigma = _f32(sigma, "sigma", self.T)
    mask = _f32(mask, "mask") if mask is not None else None
    N = x.numel() // self.T
    if pred.numel() != N * S * self.T or sigma.numel() != pred.numel():
        raise ValueError("pred/sigma must hold N*S rows")
    out = torch.empty(N * S, dtype=torch.float32, device=x.device)
    _lib.check(self.lib.qbold_nll_fwd(self.handle, _ptr(x), _ptr(mask), _ptr(pred), _ptr(sigma),
                                      _ptr(out), N, int(S), _stream()), "qbold_nll_fwd")
    return out


@_ctx_method
def kl_fwd(self, q, prior, K=70, zk=None, seed=1, voxel0=0):
    q = _f32(q, "q", 5)
    prior = _f32(prior, "prior", 5)
    N = q.numel() // 5
    zk = _f32(zk, "zk", 2) if zk is not None else None
    if zk is not None and zk.numel() != N * K * 2:
        raise ValueError("zk must be [N, K, 2]")
    out = torch.empty(q.shape[:-1], dtype=torch.float32, device=q.device)
    _lib.check(self.lib.qbold_kl_fwd(self.handle, _ptr(q), _ptr(prior), _ptr(zk), int(K), int(seed),
                                     int(voxel0), _ptr(out), N, _stream()), "qbold_kl_fwd")
    return out


@_ctx_method
def kl_closed(self, q, prior):
    q = _f32(q, "q", 5)
    prior = _f32(prior, "prior", 5)
    out = torch.empty(q.shape[:-1], dtype=torch.float32, device=q.device)
    _lib.check(self.lib.qbold_kl_closed(self.handle, _ptr(q), _ptr(prior), _ptr(out), q.numel() // 5,
                                        _stream()), "qbold_kl_closed")
    return out


@_ctx_method
def normals(self, N, n, stream_id=0, seed=1, voxel0=0):
    z = torch.empty((N, n, 2), dtype=torch.float32, device=self.device)
    _lib.check(self.lib.qbold_normals(self.handle, int(seed), int(stream_id), int(voxel0), int(n),
                                      _ptr(z), int(N), _stream()), "qbold_normals")
    return z


@_ctx_method
def add_noise(self, signal, norm_snr, snr_lo=50.0, snr_hi=120.0, seed=1, voxel0=0):
    """In place; signal [V, T] float32 contiguous cuda tensor."""
    if not (signal.is_cuda and signal.dtype == torch.float32 and signal.is_contiguous()):
        raise QboldError("add_noise needs a contiguous float32 cuda tensor (it is modified in place)")
    norm_snr = np.ascontiguousarray(norm_snr, np.float32)
    if norm_snr.size != self.T:
        raise ValueError("norm_snr must have T entries")
    ws = torch.empty(int(self.lib.qbold_noise_workspace_bytes(self.handle)), dtype=torch.uint8,
                     device=signal.device)
    _lib.check(self.lib.qbold_signal_add_noise(self.handle, _ptr(signal),
                                               norm_snr.ctypes.data_as(C.c_void_p), float(snr_lo),
                                               float(snr_hi), int(seed), int(voxel0), _ptr(ws),
                                               signal.numel() // self.T, _stream()),
               "qbold_signal_add_noise")
    return signal


@_ctx_method
def elbo_bwd(self, x, mask, q, prior, log_sigma, S=1, K=70, seed=1, voxel0=0):
    """Head gradients of the per-voxel negative ELBO (unnormalised by sum(mask)).
    Returns (sums, g_q [N,5], g_log_sigma [N,T], nll_kl [N,2])."""
    x = _f32(x, "x", self.T)
    N = x.numel() // self.T
    q = _f32(q, "q", 5)
    prior = _f32(prior, "prior", 5)
    ls = _f32(log_sigma, "log_sigma", self.T)
    mask = _f32(mask, "mask") if mask is not None else None
    sums = torch.empty(3, dtype=torch.float64, device=x.device)
    gq = torch.empty((N, 5), dtype=torch.float32, device=x.device)
    gls = torch.empty((N, self.T), dtype=torch.float32, device=x.device)
    nk = torch.empty((N, 2), dtype=torch.float32, device=x.device)
    _lib.check(self.lib.qbold_elbo_bwd(self.handle, _ptr(x), _ptr(mask), _ptr(q), _ptr(prior), _ptr(ls),
                                       int(S), int(K), int(seed), int(voxel0), _ptr(gq), _ptr(gls),
                                       _ptr(nk), _ptr(sums), _ptr(self._workspace()), N, _stream()),
               "qbold_elbo_bwd")
    return sums, gq, gls, nk


class TrainState:
    """Flat AdamW state of one encoder + the activation workspace of the training kernels."""

    def __init__(self, ctx, weights, optimiser_state=True):
        self.ctx = ctx
        self.weights = weights
        n = weights.num_params if optimiser_state else 1
        dev = weights.flat.device
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step = 0
        self.training = bool(optimiser_state)   # a training state runs Dropout (model.py:136); inference is the identity
        self.dropout_base = 0x5eed0000          # the per-step dropout seed is dropout_base + step + 1
        self._ws = None
        self._ws_n = 0
        # stream-2 voxel batches of the LDS-resident shapes: forward with saved activations in one launch
        # (False: the layer-wise exact-f32 GEMM forward, kept for comparison and for every other shape)
        self.fused_forward = True

    def workspace(self, N):
        if self._ws is None or self._ws_n < N:
            n = int(self.ctx.lib.qbold_train_workspace_floats(C.byref(self.weights.shape), int(N)))
            self._ws = torch.empty(n, dtype=torch.float32, device=self.weights.flat.device)
            self._ws_n = N
        return self._ws

    def forward(self, x, stream_sel):
        """Training forward (activations saved).  Returns (q [N,5], log_sigma [N,T] or None)."""
        ctx = self.ctx
        x = _f32(x, "x", ctx.T)
        N = x.numel() // ctx.T
        q = torch.empty((N, 5), dtype=torch.float32, device=x.device)
        ls = torch.empty((N, ctx.T), dtype=torch.float32, device=x.device) if stream_sel == 2 else None
        sh = self.weights.shape
        self._set_dropout_seed()
        if (stream_sel == 2 and self.fused_forward and Context.fits_fused(sh) and sh.channelwise_gating
                and sh.precision == 0 and 0 < N < (1 << 23) and not (self.training and sh.dropout_rate > 0.0)):
            # one launch, every saved tensor written once (encoder_kernels.hip, encoder_train_fwd_kernel); the two
            # tensors per block that the backward will recompute (it says which) are left out
            save_all = 2 - max(0, min(2, int(ctx.lib.qbold_encoder_train_bwd_recomputes(ctx.handle, C.byref(sh), N))))
            _lib.check(ctx.lib.qbold_encoder_train_fwd_fused(ctx.handle, C.byref(sh), self.weights.packed_ptr(),
                                                             _ptr(x), save_all, _ptr(self.workspace(N)), _ptr(q),
                                                             _ptr(ls), N, _stream()), "qbold_encoder_train_fwd_fused")
        else:
            _lib.check(ctx.lib.qbold_encoder_train_fwd(ctx.handle, C.byref(sh),
                                                       _ptr(self.weights.flat), _ptr(x), int(stream_sel),
                                                       _ptr(self.workspace(N)), _ptr(q), _ptr(ls), N,
                                                       _stream()), "qbold_encoder_train_fwd")
        self._n = N
        return q, ls

    def _set_dropout_seed(self):
        """Keras runs Dropout in training only: a training state's forward draws the step's mask (and its backward
        regenerates it from the same seed); any other forward sees seed 0, the identity."""
        sh = self.weights.shape
        sh.dropout_seed = (self.dropout_base + self.step + 1) if (self.training and sh.dropout_rate > 0.0) else 0

    def backward(self, stream_sel, g_q, g_ls=None, sums=None):
        """Fills self.grad (canonical layout) from the head gradients of the last forward()."""
        ctx = self.ctx
        self._set_dropout_seed()
        _lib.check(ctx.lib.qbold_encoder_train_bwd(ctx.handle, C.byref(self.weights.shape),
                                                   _ptr(self.weights.flat), int(stream_sel),
                                                   _ptr(self._ws), _ptr(g_q), _ptr(g_ls), _ptr(sums),
                                                   _ptr(self.grad), self._n, _stream()),
                   "qbold_encoder_train_bwd")
        return self.grad

    def forward_spatial(self, x5):
        """Stream 2 on an image-crop batch x5 [B, X, Y, Z, T] with the 3x3x1 convolutions.
        Returns (q [V,5], log_sigma [V,T]) with V = B X Y Z."""
        ctx = self.ctx
        x = _f32(x5, "x", ctx.T)
        if x.dim() != 5:
            raise ValueError("forward_spatial expects [B, X, Y, Z, T]")
        B, X, Y, Z, _ = x.shape
        self._geom = Geometry(B, X, Y, Z)
        self._set_dropout_seed()
        N = B * X * Y * Z
        q = torch.empty((N, 5), dtype=torch.float32, device=x.device)
        ls = torch.empty((N, ctx.T), dtype=torch.float32, device=x.device)
        _lib.check(ctx.lib.qbold_encoder_spatial_fwd(ctx.handle, C.byref(self.weights.shape),
                                                     _ptr(self.weights.flat), _ptr(x), C.byref(self._geom),
                                                     _ptr(self.workspace(N)), _ptr(q), _ptr(ls), _stream()),
                   "qbold_encoder_spatial_fwd")
        self._n = N
        return q, ls

    def backward_spatial(self, g_q, g_ls, sums=None):
        ctx = self.ctx
        self._set_dropout_seed()
        _lib.check(ctx.lib.qbold_encoder_spatial_bwd(ctx.handle, C.byref(self.weights.shape),
                                                     _ptr(self.weights.flat), C.byref(self._geom),
                                                     _ptr(self._ws), _ptr(g_q), _ptr(g_ls), _ptr(sums),
                                                     _ptr(self.grad), _stream()),
                   "qbold_encoder_spatial_bwd")
        return self.grad

    def synth_loss_bwd(self, y_true, q, inv_gamma_alpha=0.0, inv_gamma_beta=0.0):
        """Pre-training loss per voxel and its head gradient (already divided by N); optional
        inverse-gamma prior on the marginal variances (model.py:492-507)."""
        ctx = self.ctx
        y = _f32(y_true, "y_true")
        ld = y.shape[-1]
        N = q.shape[0]
        gq = torch.empty((N, 5), dtype=torch.float32, device=q.device)
        lv = torch.empty(N, dtype=torch.float32, device=q.device)
        _lib.check(ctx.lib.qbold_synth_loss_bwd(ctx.handle, _ptr(y), int(ld), _ptr(q), _ptr(gq), _ptr(lv),
                                                1.0 / N, float(inv_gamma_alpha), float(inv_gamma_beta), N, _stream()),
                   "qbold_synth_loss_bwd")
        return lv, gq

    def adamw(self, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-7, ranges=None):
        """One AdamW step.  ranges: [(offset, length)] of the canonical blob that received a gradient in this
        phase (EncoderWeights.param_ranges); parameters outside them are not touched -- neither the Adam update
        nor the weight decay -- as Keras / tfa leave variables whose gradient is None."""
        self.step += 1
        w = self.weights.flat
        for off, n in (ranges if ranges is not None else [(0, self.weights.num_params)]):
            _lib.check(self.ctx.lib.qbold_adamw_step(self.ctx.handle, _ptr(w[off:off + n]), _ptr(self.grad[off:off + n]),
                                                     _ptr(self.m[off:off + n]), _ptr(self.v[off:off + n]), int(n),
                                                     float(lr), float(beta1), float(beta2), float(eps),
                                                     float(weight_decay), self.step, _stream()),
                       "qbold_adamw_step")
        self.weights.mark_dirty()


@_ctx_method
def smoothness(self, q5, mask5, weight=0.0, g_q=None):
    """smoothness_loss numerator on a crop batch: q5 [B,X,Y,Z,5], mask5 [B,X,Y,Z(,1)].  Returns the
    device double sum |dx| + |dy|; if g_q [V,5] is given, weight * d sum / d q is added to it."""
    q = _f32(q5, "q", 5)
    if q.dim() != 5:
        raise ValueError("smoothness expects [B, X, Y, Z, 5]")
    B, X, Y, Z, _ = q.shape
    m = _f32(mask5, "mask").reshape(-1)
    if m.numel() != B * X * Y * Z:
        raise ValueError("mask does not match q")
    geom = Geometry(B, X, Y, Z)
    tv = torch.empty(1, dtype=torch.float64, device=q.device)
    _lib.check(self.lib.qbold_smoothness(self.handle, _ptr(q), _ptr(m), C.byref(geom), float(weight),
                                         _ptr(g_q), _ptr(tv), _stream()), "qbold_smoothness")
    return tv
