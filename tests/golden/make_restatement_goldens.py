"""Writes tests/golden/restatement_goldens.npz from the CPU oracle (oracle/qbold_oracle.c).

THESE ARE RESTATEMENT GOLDENS, NOT REFERENCE OUTPUTS: TensorFlow cannot be imported in the build
container, the reference ships no tests or vectors, so the only fixture produced by
reference code is merged_config_optimal.json.  The goldens pin the oracle against drift and let the
GPU tests run against committed numbers as well as against the live oracle.
"""
import configparser
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, fit_wls, init_weights, synth_inputs  # noqa: E402

cfg = configparser.ConfigParser()
cfg.read(os.path.join(ROOT, "config"))
params = dict(cfg["DEFAULT"])
o32, o64 = Oracle("f32", params), Oracle("f64", params)

out = {}
out["F_x"] = np.array([0, .05, .1, .5, 1, 2, 4, 8, 16], np.float64)
out["F_f32"] = o32.tissue_F(out["F_x"])
out["F_f64"] = o64.tissue_F(out["F_x"])
pts = np.array([[0.4, 0.12], [0.4, 0.025], [0.04, 0.001], [0.84, 0.201]], np.float32)
out["sig_pts"] = pts
out["sig_f32"] = o32.signal_fwd(pts)
out["sig_f64"] = o64.signal_fwd(pts)
out["taus"] = o32.taus

n, S, K = 256, 4, 70
w = init_weights(T=11, U=60, L=2, seed=1)
w["gate_offset"] = -3.0
x, y = synth_inputs(n, params, seed=1, oracle=o32)
prior, q, sigma = o32.encoder_fwd(w, x)
rng = np.random.default_rng(1)  # seeded NumPy PCG64
zs = rng.standard_normal((n, S, 2)).astype(np.float32)
zk = rng.standard_normal((n, K, 2)).astype(np.float32)
mask = (rng.uniform(size=n) > 0.25).astype(np.float32)
e = o32.elbo(x, mask, q, prior, sigma, zs, zk)
zm = rng.standard_normal((n, 20, 2)).astype(np.float32)
means, var = o32.moments(q, zm)
for k in ("W0", "b0", "Wc", "bc", "Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Wf", "bf", "Ws", "bs"):
    out["w_" + k] = w[k]
out.update(x=x, truth=y, prior=prior, q=q, sigma=sigma, zs=zs, zk=zk, mask=mask, nll_v=e["nll_v"],
           kl_v=e["kl_v"], sums=e["sums"], elbo=np.float64(e["elbo"]), zm=zm, means=means, vars=var,
           philox_z=o32.philox_normals(1, 0, 5, 8, 6))
# round-1 widening rows: image crops (3x3x1 context + TV), signal-model options, log-linear WLS
w9 = init_weights(T=11, U=12, L=2, seed=2, taps=9)
w9["gate_offset"] = -1.0
crop = x[:2 * 5 * 4 * 3].reshape(2, 5, 4, 3, 11)
sp_q, sp_sigma = o32.encoder_fwd_spatial(w9, crop)
crop_mask = (rng.uniform(size=(2, 5, 4, 3)) > 0.2).astype(np.float32)
for k in ("W0", "b0", "Wc", "bc", "Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Wf", "bf", "Ws", "bs"):
    out["w9_" + k] = w9[k]
out.update(crop=crop, crop_mask=crop_mask, sp_q=sp_q, sp_sigma=sp_sigma,
           tv=np.float64(o32.smoothness_loss(sp_q, crop_mask)))
hct = rng.uniform(0.25, 0.5, 16).astype(np.float32)
alt = np.stack([rng.uniform(0.05, 0.8, 16), rng.uniform(0.002, 0.3, 16)], -1).astype(np.float32)
idx = rng.integers(4, 12, 16).astype(np.int32)
out.update(ex_y=y[:16], ex_hct=hct, ex_alt=alt, ex_idx=idx,
           ex_sig=o32.signal_fwd_ex(y[:16], hct=hct, alt=alt, from_idx=idx))
wo, wd, wr = fit_wls(x[:32].astype(np.float64) * 200.0, params)
out.update(wls_oef=wo, wls_dbv=wd, wls_r2p=wr)
np.savez_compressed(os.path.join(HERE, "restatement_goldens.npz"), **out)
print("wrote", len(out), "arrays; elbo", e["elbo"])
