"""Forward signal model and synthetic data -- the host mirror of the reference's signals.py.

Same class / function names, argument order and defaults as the reference
(SignalGenerationLayer signals.py:13-248, create_synthetic_dataset :251-300), operating on
float32 ROCm tensors.  All arithmetic on the signals is done by libqbold_hip.so
(qbold_signal_fwd / qbold_signal_add_noise); torch is used for allocation, reshaping, meshgrid
and shuffling only.
"""
import math

import numpy as np
import torch

from .ops import Context

# normalised per-tau SNR of real data, relative to the tau = 0 image (signals.py:117-121)
_NORM_SNR_11 = np.array([0.985, 1.00, 1.01, 1., 0.97, 0.95, 0.93, 0.90, 0.86, 0.83, 0.79],
                        dtype=np.float32)


def _norm_snr(T):
    if T == 11:
        return _NORM_SNR_11
    if T == 24:
        return (1.0 - (np.abs(np.arange(-0.028, 0.065, 0.004)) * 3.0)).astype(np.float32)
    # the reference raises NameError here (signals.py:117-124 defines norm_snr for 11 / 24 only)
    raise ValueError(f"the reference defines the noise model for 11 or 24 taus only, got {T}")


class SignalGenerationLayer:
    """Encapsulates the qBOLD signal equations (signals.py:13-140).

    layer(x) with x[..., 2] = (OEF, DBV) returns the predicted ASE signal [..., T]."""

    def __init__(self, system_parameters, full_model, include_blood, misaligned_prob=0.0,
                 variable_hct=False, device=None, seed=1):
        p = system_parameters
        self._gamma = float(p['gamma'])
        self._b0 = float(p['b0'])
        self._dchi = float(p['dchi'])
        self._te = float(p['te'])
        self._r2t = float(p['r2t'])
        self._tr = float(p['tr'])
        self._ti = float(p['ti'])
        self._t1b = float(p['t1b'])
        self._simulate_noise = p['simulate_noise'] == 'True'
        self._weighted_noise = p['tau_weighted'] == 'True'
        self._snr = int(p['snr'])
        if not variable_hct:  # signals.py:45-46
            self.hct = float(p['hct'])
        # full_model / include_blood arrive as the strings 'True' / 'False' from the reference's
        # __main__ (signals.py:330); any non-empty string is truthy there as well
        self._full_model = bool(full_model)
        self._include_blood = bool(include_blood)
        self._misaligned_prob = misaligned_prob
        self._variable_hct = variable_hct
        self._ctx = Context(p, self._full_model, self._include_blood, device=device)
        self._taus = torch.as_tensor(self._ctx.taus)
        self._seed = int(seed)
        self._calls = 0

    @property
    def context(self):
        return self._ctx

    def __call__(self, input, *args, **kwargs):
        return self.call(input, *args, **kwargs)

    def _misalignment(self, y):
        """The random draws of signals.py:80-93 (torch generator on the device; every call fresh):
        -> (alt [V,2], from_index [V] int32 with T for the aligned voxels)."""
        T, V = self._ctx.T, y.shape[0]
        g = torch.Generator(device=y.device)
        g.manual_seed(self._seed + 104723 * (self._calls + 1))
        misaligned = torch.rand(V, generator=g, device=y.device) < self._misaligned_prob
        from_index = torch.randint(4, T - 1, (V,), generator=g, device=y.device, dtype=torch.int32)
        from_index = torch.where(misaligned, from_index, torch.full_like(from_index, T))
        noise = torch.randn(V, 2, generator=g, device=y.device)
        alt = torch.stack([torch.clamp(noise[:, 0] * 0.15 + y[:, 0], 0.05, 0.8),
                           torch.clamp(noise[:, 1] * 0.05 + y[:, 1], 0.002, 0.3)], -1)
        return alt, from_index

    def call(self, input, *args, **kwargs):
        if self._variable_hct:
            if input.shape[-1] != 3:
                raise AssertionError('Input should have 3 elements in last dimension, OEF, DBV and hct')
        elif input.shape[-1] != 2:
            raise AssertionError('Input should have 2 elements in last dimension, OEF and DBV')
        if self._variable_hct or self._misaligned_prob > 0.0:
            flat = input.reshape(-1, input.shape[-1])
            y = flat[:, :2].contiguous()
            hct = flat[:, 2].contiguous() if self._variable_hct else None
            alt, from_index = self._misalignment(y) if self._misaligned_prob > 0.0 else (None, None)
            signal = self._ctx.signal_fwd_ex(y, hct, alt, from_index).reshape(
                tuple(input.shape[:-1]) + (self._ctx.T,))
        else:
            signal = self._ctx.signal_fwd(input)
        if self._simulate_noise:
            flat = signal.reshape(-1, signal.shape[-1])
            # every call draws fresh noise, as tf.random does (signals.py:124,128)
            self._ctx.add_noise(flat, _norm_snr(flat.shape[-1]), 50.0, 120.0,
                                seed=self._seed + 7919 * self._calls)
            self._calls += 1
        return signal

    def gradient(self, input, grad_output):
        """Vector-Jacobian product of call() (noise-free part), what tf.GradientTape returns."""
        if self._variable_hct or self._misaligned_prob > 0.0:
            raise NotImplementedError("gradients through variable_hct / misalignment: these options only "
                                      "occur in synthetic-data generation (signals.py:251-300)")
        return self._ctx.signal_bwd(input, grad_output)

    @staticmethod
    def calculate_dw_static(oef, hct, gamma, b0, dchi):
        return (4.0 / 3.0) * math.pi * gamma * b0 * dchi * hct * oef

    def calculate_dw(self, oef, hct):
        return SignalGenerationLayer.calculate_dw_static(oef, hct, self._gamma, self._b0, self._dchi)

    def calculate_r2p(self, oef, dbv, hct):
        return self.calculate_dw(oef, hct) * dbv


def _truncated_normal(rng, n, loc, scale, low, high):
    out = rng.standard_normal(n) * scale + loc
    bad = (out < low) | (out > high)
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum())) * scale + loc
        bad = (out < low) | (out > high)
    return out


def create_synthetic_dataset(params, full_model, use_blood, misaligned_prob, variable_hct=False,
                             uniform_prop=0.1, device=None, seed=1, sample_size=None):
    """signals.create_synthetic_dataset (signals.py:251-300): sample_size OEF values x sample_size
    DBV values on a meshgrid, shuffled, pushed through the (noisy) forward model in 10 chunks.
    Returns (x [N, T], y [N, 3] = (OEF, DBV, R2')) as ROCm tensors.  sample_size overrides the INI
    value (2500 -> 6.25 M voxels) for smaller runs."""
    n = int(params['sample_size']) if sample_size is None else int(sample_size)
    rng = np.random.default_rng(seed)
    n_u, n_n = round(n * uniform_prop), round(n * (1.0 - uniform_prop))
    oefs = np.concatenate([
        rng.uniform(float(params['oef_start']), float(params['oef_end']), n_u),
        np.clip(rng.standard_normal(n_n) * float(params['oef_std']) + float(params['oef_mean']),
                float(params['oef_start']), float(params['oef_end']))])
    dbvs = np.concatenate([
        rng.uniform(float(params['dbv_start']), float(params['dbv_end']), n_u),
        _truncated_normal(rng, n_n, float(params['dbv_mean']), float(params['dbv_std']),
                          float(params['dbv_start']), float(params['dbv_end']))])
    sig_layer = SignalGenerationLayer(params, full_model, use_blood, misaligned_prob=misaligned_prob,
                                      variable_hct=variable_hct, device=device, seed=seed)
    dev = sig_layer.context.device
    xx, yy = torch.meshgrid(torch.as_tensor(oefs, dtype=torch.float32, device=dev),
                            torch.as_tensor(dbvs, dtype=torch.float32, device=dev), indexing='ij')
    train_y = torch.stack([xx.reshape(-1), yy.reshape(-1)], dim=1)
    if variable_hct:  # tf.random.uniform(minval=0.34, maxval=0.34): a constant column (signals.py:273-276)
        train_y = torch.cat([train_y, torch.full_like(train_y[:, :1], 0.34)], dim=-1)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    train_y = train_y[torch.randperm(train_y.shape[0], generator=g, device=dev)]  # tf.random.shuffle
    chunk = train_y.shape[0] // 10  # "break into chunks" -- the noise std uses per-chunk means
    train_x = torch.cat([sig_layer(train_y[i * chunk:(i + 1) * chunk].contiguous()) for i in range(10)])
    train_y = train_y[:train_x.shape[0]]
    hct = train_y[:, 2] if variable_hct else sig_layer.hct
    r2p = sig_layer.calculate_r2p(train_y[:, 0], train_y[:, 1], hct)
    return train_x, torch.cat([train_y[:, :2], r2p[:, None]], dim=-1)
