"""Second, independently written CPU restatement of the hot path -- torch float32 on the host, at the
REFERENCE'S OP GRANULARITY (whole-batch tensor ops, the materialised [V, T, 129] Bessel tensor of
signals.py:168-171, the S-fold tiled batch of model.py:245-246), where oracle/qbold_oracle.c is a scalar loop
per voxel.

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference's TensorFlow cannot run here; see
oracle/qbold_oracle.h).  Two uses: (1) tests/test_oracle.py checks that this restatement and the C one agree to
float32 rounding on BASELINE config 1 -- two restatements written separately from the same source lines;
(2) bench.py's cpu_baseline times it as "the reference's CPU path at the reference's op granularity".
The product package never imports this module.

Every function cites the reference lines it follows.  The Bessel function is Cephes' single-precision j0f
(what Eigen's generic_j0<float> behind tf.math.special.bessel_j0 evaluates), written out in torch ops because
torch.special.bessel_j0 uses the double-precision Cephes scheme in float32: at Simpson node 0 (u = 1e-5) that
returns 1 - O(1e-8) instead of exactly 1, and (1 - J0) / (3 u^2) turns the difference into an O(1) error of the
integral.
"""
import math

import numpy as np
import torch

F32 = torch.float32


def _poly(z, coef):
    acc = torch.full_like(z, coef[0])
    for c in coef[1:]:
        acc = acc * z + c
    return acc


_JP = (-6.068350350393235E-008, 6.388945720783375E-006, -3.969646342510940E-004, 1.332913422519003E-002,
       -1.729150680240724E-001)
_MO = (-6.838999669318810E-002, 1.864949361379502E-001, -2.145007480346739E-001, 1.197549369473540E-001,
       -3.560281861530129E-003, -4.969382655296620E-002, -3.355424622293709E-006, 7.978845717621440E-001)
_PH = (3.242077816988247E+001, -3.630592630518434E+001, 1.756221482109099E+001, -4.974978466280903E+000,
       1.001973420681837E+000, -1.939906941791308E-001, 6.490598792654666E-002, -1.249992184872738E-001)


def j0f(x):
    """Cephes j0f.c: |x| <= 2: (z - DR1) P(z) (1 - z/4 below 1e-3); else sqrt(1/x) MO(1/x) cos(x + PH - pi/4)."""
    x = x.abs()
    z = x * x
    small = torch.where(x < 1.0e-3, 1.0 - 0.25 * z, (z - 5.78318596294678452118) * _poly(z, _JP))
    q = 1.0 / x.clamp_min(1e-30)
    w = torch.sqrt(q)
    p = w * _poly(q, _MO)
    xn = q * _poly(q * q, _PH) - 0.7853981633974483096
    large = p * torch.cos(xn + x)
    return torch.where(x <= 2.0, small, large)


def tau_grid(params):
    """tf.range(tau_start, tau_end, tau_step, float32) (signals.py:34-35): start + i * step in float32."""
    ts, te, st = (np.float32(float(params[k])) for k in ("tau_start", "tau_end", "tau_step"))
    n = int(math.ceil(abs(float(te) - float(ts)) / abs(float(st))))
    return torch.tensor([ts + np.float32(i) * st for i in range(n)], dtype=F32)


def signal_model(oef_dbv, params, full_model=True, include_blood=True, chunk=4096):
    """SignalGenerationLayer.call without noise (signals.py:55-114) with calc_tissue (:152-209) and calc_blood
    (:233-247).  oef_dbv [V, 2] -> [V, T].  Evaluated in chunks of voxels like create_synthetic_dataset's loop
    (signals.py:281-285): each chunk materialises the [chunk, T, 129] tensor."""
    p = {k: float(params[k]) for k in ("gamma", "b0", "dchi", "te", "r2t", "tr", "ti", "t1b", "hct")}
    taus = tau_grid(params)
    y = torch.as_tensor(oef_dbv, dtype=F32).reshape(-1, 2)
    out = []
    u = torch.linspace(1e-5, 1.0, 2 ** 7 + 1, dtype=F32)                        # signals.py:166-168
    for a in range(0, y.shape[0], chunk):
        oef, dbv = y[a:a + chunk, 0:1], y[a:a + chunk, 1:2]
        dw = F32_const((4.0 / 3.0) * math.pi * p["gamma"] * p["b0"] * p["dchi"] * p["hct"]) * oef  # :142-147
        if full_model:
            arg = 1.5 * (taus[None, :] * dw)[..., None] * u                       # [V, T, 129], :170
            integrand = (2 + u) * torch.sqrt(1 - u) * (1.0 - j0f(arg)) / (3.0 * torch.square(u))  # :169-171
            ya, yb, ym = integrand[..., 0:-2:2], integrand[..., 2::2], integrand[..., 1:-1:2]     # :180-182
            h = (u[2] - u[0]) / 2.0
            integral = ((ya + yb + 4.0 * ym) * (h / 3.0)).sum(-1)                 # :183-185
            tissue = torch.exp(-dbv * integral) * math.exp(-p["te"] * p["r2t"])   # :169,172
        else:
            tc = 1.0 / dw                                                         # :189
            r2p = dw * dbv
            under = (taus.abs()[None, :] < tc).to(F32)                            # :197-201
            e = math.exp(-p["r2t"] * p["te"])
            s1 = e * torch.exp(-(0.3 * (r2p * taus[None, :]) ** 2) / dbv)         # :204
            s2 = e * torch.exp(dbv - (r2p * taus[None, :]))                       # :205
            tissue = s1 * under + s2 * (1.0 - under)
        if include_blood:
            m_bld = 1 - (2 - math.exp(-(p["tr"] - p["ti"]) / p["t1b"])) * math.exp(-p["ti"] / p["t1b"])  # :105
            bw = F32_const(m_bld * 0.775) * dbv                                   # :102,107
            r2b, td = 1.0 / 0.189, (2.6 ** 2.0) / 2.0 * 1e-3                      # :235-238
            g0 = F32_const((4 / 45) * p["hct"] * (1 - p["hct"])) * (F32_const(4.0 * math.pi * p["b0"] * p["dchi"]) * oef) ** 2
            te = p["te"]
            bracket = (te / td) + math.sqrt(0.25 + te / td) + 1.5 \
                - 2.0 * torch.sqrt(0.25 + ((te + taus) / td)) - 2.0 * torch.sqrt(0.25 + ((te - taus) / td))  # :242-247
            blood = math.exp(-r2b * te) * torch.exp(-(F32_const(0.5 * p["gamma"] ** 2) * g0 * F32_const(td ** 2)) * bracket[None, :])
        else:
            bw, blood = dbv, torch.zeros_like(tissue)                             # :99,110
        out.append((1 - bw) * tissue + bw * blood)                                # :112-114
    return torch.cat(out, 0)


def F32_const(v):
    return torch.tensor(v, dtype=F32)


def normalise(x, se_idx, multi=False):
    """EncoderTrainer.normalise_data (model.py:97-113)."""
    c = torch.clamp(torch.as_tensor(x, dtype=F32), 1e-2, 1e8)
    den = c[:, se_idx - 1:se_idx + 2].mean(-1, keepdim=True) if multi else c[:, se_idx:se_idx + 1]
    return torch.log(c / den)


def encoder(w, x, se_idx, gate_offset, multi=False):
    """create_encoder on (N,1,1,1,T) voxel batches (model.py:122-223; the 3x3x1 kernels act through their centre
    tap).  Returns (out1 [N,5], out2 [N,5], sigma [N,T])."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=F32)
    n = normalise(x, se_idx, multi)
    h = torch.relu(n @ t(w["W0"]) + t(w["b0"]))                                   # :181
    a = b = h
    L = np.asarray(w["Wc"]).shape[0]
    for l in range(L):                                                            # create_block, :142-174
        Wc, bc = t(w["Wc"][l]), t(w["bc"][l])
        ctr = lambda k: t(np.asarray(k)[l][1, 1]) if np.asarray(k).ndim == 5 else t(np.asarray(k)[l])  # centre tap
        a = torch.relu(a @ Wc + bc)                                               # :144-145
        skip = torch.relu(b @ Wc + bc)                                            # :148
        r = torch.relu(torch.relu(b) @ ctr(w["Wr1"]) + t(w["br1"][l]))            # :151-155
        r = r @ ctr(w["Wr2"]) + t(w["br2"][l])                                    # :156
        g = torch.sigmoid(r @ t(w["Wg"][l]) + t(w["bg"][l]) + gate_offset)        # :164,169
        b = skip * (1.0 - g) + r * g                                              # :170
    Wf, bf = t(w["Wf"]), t(w["bf"])
    return a @ Wf + bf, b @ Wf + bf, torch.exp(b @ t(w["Ws"]) + t(w["bs"]))      # :196-214


def _std(p):
    return torch.tanh(p) * 3.0 - 1.0                                              # transform_std, model.py:288-290


def _offdiag(p):
    return torch.tanh(p) * math.exp(-2.0)                                         # transform_offdiag, :292-294


def reparam(q, z):
    """ReparamTrickLayer.call + forward_transform (model.py:15-50, 299-305): q [R,5], z [R,2] -> (OEF, DBV) [R,2]."""
    a = q[:, 0] + z[:, 0] * torch.exp(_std(q[:, 1]))
    b = q[:, 2] + z[:, 0] * _offdiag(q[:, 4]) + z[:, 1] * torch.exp(_std(q[:, 3]))
    return torch.stack([torch.sigmoid(a) * 0.8 + 0.04, torch.sigmoid(b) * 0.2 + 0.001], -1)


def neg_log_prob(y, p):
    """logit_gaussian_mvg_log_prob (model.py:376-400) with squared_whitened_residual (:423-441) and
    calculate_log_chol_det (:443-447): NEGATIVE log density of y [R,2] under p [R,5]."""
    so, sd, c = _std(p[:, 1]), _std(p[:, 3]), _offdiag(p[:, 4])
    x = torch.stack([(y[:, 0] - 0.04) / 0.8, (y[:, 1] - 0.001) / 0.2], -1)        # backwards_transform, :307-311
    x = torch.clamp(x, 1e-6, 1.0 - 1e-6)                                           # :394-395
    lg = torch.log(x / (1.0 - x))                                                  # logit, :10-12
    r0, r1 = lg[:, 0] - p[:, 0], lg[:, 1] - p[:, 2]
    w0 = r0 * torch.exp(-so)
    w1 = r1 * torch.exp(-sd) + r0 * (torch.exp(-so - sd) * c * -1.0)
    swr = w0 * w0 + w1 * w1
    nll = math.log(2.0 * math.pi) + 0.5 * (2.0 * (so + sd)) + 0.5 * swr            # :385-390
    return nll + (torch.log(x) + torch.log(1.0 - x)).sum(-1)                       # :398


def elbo(x, mask, q, prior, sigma, zs, zk, params, se_idx, kl_tiled=False):
    """build_fine_tuner's sampling (model.py:245-248), fine_tune_loss_fn (:527-568; Gaussian, linear data,
    one-image normalisation) and kl_loss -> mvg_kl_samples (:654-665, 592-610) on a voxel batch, tiled S-fold by
    concatenation as the reference does.  zs [N,S,2], zk [N,K,2] (kl_tiled: [N,S*K,2] -- the reference's own
    draw count, K per tiled copy).  Returns dict(nll_v [N], kl_v [N], nll, kl, elbo)."""
    x = torch.as_tensor(x, dtype=F32)
    N, T = x.shape
    mask = torch.as_tensor(mask, dtype=F32).reshape(N)
    q, prior, sigma = (torch.as_tensor(a, dtype=F32) for a in (q, prior, sigma))
    zs = torch.as_tensor(zs, dtype=F32)
    zk = torch.as_tensor(zk, dtype=F32)
    S = zs.shape[1]
    tile = lambda a: torch.cat([a for _ in range(S)], 0)                           # :245-246, 529
    qt, st, xt, mt = tile(q), tile(sigma), tile(x), tile(mask)
    z = zs.permute(1, 0, 2).reshape(S * N, 2)                                      # copy s of voxel n takes draw zs[n, s]
    pred = signal_model(reparam(qt, z), params)                                    # :248,273
    yt = xt / (xt[:, se_idx:se_idx + 1] + 1e-3)                                    # :544
    yp = pred / (pred[:, se_idx:se_idx + 1] + 1e-3)                                # :545
    res = yt - yp
    nll_rows = (torch.log(st) + math.log(math.sqrt(2.0 * math.pi)) + 0.5 * torch.square(res / st)).sum(-1)  # :561-563
    nll_v = nll_rows.reshape(S, N).mean(0)
    nll = (nll_rows * mt).sum() / mt.sum()                                         # :564-566
    # KL: log q - log p at K reparameterised draws of q (q stop-gradient inside log q; no gradients here)
    if kl_tiled:
        K = zk.shape[1] // S
        zkt = zk.reshape(N, S, K, 2).permute(1, 0, 2, 3).reshape(S * N, K, 2)
        qk, pk = qt, tile(prior)
    else:
        K = zk.shape[1]
        zkt, qk, pk = zk, q, prior
    acc = torch.zeros(qk.shape[0], dtype=F32)
    for k in range(K):                                                             # :596-597
        yk = reparam(qk, zkt[:, k])
        acc = acc + (neg_log_prob(yk, pk) - neg_log_prob(yk, qk))                  # log q - log p = nlp_p - nlp_q
    kl_rows = acc / K                                                              # :607
    if kl_tiled:
        kl_v = kl_rows.reshape(S, N).mean(0)
        mk = mt
    else:
        kl_v, mk = kl_rows, mask
    kl = torch.where(mk > 0, kl_rows, torch.zeros_like(kl_rows)).sum() / mk.sum()  # :661-663
    return dict(nll_v=nll_v.numpy(), kl_v=kl_v.numpy(), nll=float(nll), kl=float(kl), elbo=float(nll + kl))
