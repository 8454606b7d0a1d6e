#!/usr/bin/env python3
"""Writes a stand-in for the reference's (private) real-data directory: ASE_scan.npy, ASE_INF.npy,
ASE_SUP.npy, hyperv_ase.npy, baseline_ase.npy shaped [subj, X, Y, 8, T+2] = signals + grey-matter mask
+ brain mask (train.py:208-221, data_preprocessing.py:265-266), from smooth random OEF / DBV maps pushed
through the forward model with the reference's noise model.  Usage:
    python scripts/make_synthetic_volumes.py OUT_DIR [--subjects 4 --size 96]
    python train.py configurations/optimal.yaml -d OUT_DIR
"""
import argparse
import configparser
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from qbold_vi_amd.signals import SignalGenerationLayer  # noqa: E402


def smooth_field(rng, shape, lo, hi, scale):
    from scipy.ndimage import gaussian_filter
    f = gaussian_filter(rng.standard_normal(shape), sigma=(0, scale, scale, 0.7))
    f = (f - f.min()) / (f.max() - f.min() + 1e-12)
    return lo + (hi - lo) * f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--subjects", type=int, default=4)
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "config"))
    params = dict(cfg["DEFAULT"], simulate_noise="True")
    layer = SignalGenerationLayer(params, True, True)
    rng = np.random.default_rng(a.seed)
    os.makedirs(a.out, exist_ok=True)
    S, X, Z = a.subjects, a.size, 8
    xx, yy = np.meshgrid(np.linspace(-1, 1, X), np.linspace(-1, 1, X), indexing="ij")
    brain = ((xx / 0.85) ** 2 + (yy / 0.7) ** 2 < 1.0)[None, :, :, None] * np.ones((S, 1, 1, Z))
    for name in ("ASE_scan", "ASE_INF", "ASE_SUP", "hyperv_ase", "baseline_ase"):
        oef = smooth_field(rng, (S, X, X, Z), 0.2, 0.6, 6.0)
        dbv = smooth_field(rng, (S, X, X, Z), 0.01, 0.08, 4.0)
        y = torch.as_tensor(np.stack([oef, dbv], -1).reshape(-1, 2), dtype=torch.float32, device="cuda")
        sig = layer(y).cpu().numpy().reshape(S, X, X, Z, -1) * 100.0
        gm = brain * (smooth_field(rng, (S, X, X, Z), 0, 1, 3.0) > 0.45)
        vol = np.concatenate([sig * brain[..., None], gm[..., None], brain[..., None]], -1).astype(np.float32)
        np.save(os.path.join(a.out, name + ".npy"), vol)
        print(name, vol.shape)


if __name__ == "__main__":
    main()
