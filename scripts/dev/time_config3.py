"""Dev timing of the config-3 step (one-launch wide encoder, then the log-sigma ELBO kernel) issued whole or in
voxel chunks, HIP events around the C entry points."""
import configparser, ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qbold_vi_amd import _lib
from qbold_vi_amd.ops import Context, EncoderWeights, _ptr, _stream
from qbold_vi_amd.init import init_encoder_weights
cfg = configparser.ConfigParser(); cfg.read(os.path.join(ROOT, "config")); p = dict(cfg["DEFAULT"])
p.update(tau_start="-0.015", tau_end="0.065", tau_step="0.00125")
ctx = Context(p, True, True)
n = 1 << 20
T, S, K = 64, 32, 70   # bench.py --config 3 defaults
w = init_encoder_weights(T=T, U=256, L=2, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=1)
ew = EncoderWeights(ctx, T, 256, 2, True, -3.0).set_from_arrays(w)
x = torch.rand((n, T), device="cuda") * 0.5 + 0.2
mask = torch.ones(n, device="cuda")
prior = torch.tensor([0.0, -0.5, -1.0, -0.5, 0.0], device="cuda").repeat(n, 1).contiguous()
q = torch.empty((n, 5), device="cuda"); ls = torch.empty((n, T), device="cuda")
nk = torch.empty((n, 2), device="cuda")
sums = torch.empty((16, 3), dtype=torch.float64, device="cuda")
ws = ctx._workspace()
lib = _lib.load()
def enc(a, b):
    _lib.check(lib.qbold_encoder_fused_fwd(ctx.handle, C.byref(ew.shape), ew.fused_ptr(), _ptr(x[a:b]), _ptr(q[a:b]), _ptr(ls[a:b]), b - a, _stream()), "fused")
def elbo(a, b, c):
    _lib.check(lib.qbold_elbo_fwd_logsigma(ctx.handle, _ptr(x[a:b]), _ptr(mask[a:b]), _ptr(q[a:b]), _ptr(prior[a:b]), _ptr(ls[a:b]),
                                           S, K, 1, a, _ptr(nk[a:b]), _ptr(sums[c]), _ptr(ws), b - a, _stream()), "elbo")
def timeit(f, reps=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print(f"encoder alone {timeit(lambda: enc(0, n)):.4f} ms   elbo alone {timeit(lambda: elbo(0, n, 0)):.4f} ms")
for sel in [int(a) for a in sys.argv[1:] if a.isdigit()]:   # e.g. 8 = QBOLD_KSEL_X_TABLE: the x-indexed ELBO kernel
    ctx.set_kernel_selection(sel)
    print(f"kernel selection {sel}: elbo alone {timeit(lambda: elbo(0, n, 0)):.4f} ms")
    ctx.set_kernel_selection(0)
for chunks in ((1,) if len(sys.argv) > 1 else (1, 2, 4, 8, 16)):
    m = n // chunks
    def both():
        for c in range(chunks):
            enc(c * m, (c + 1) * m)
            elbo(c * m, (c + 1) * m, c)
    print(f"chunks {chunks:2d}: step {timeit(both):.4f} ms")
