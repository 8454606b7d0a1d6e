"""Dev timing of the one-launch wide encoder alone (config-3 shapes), HIP events around the C entry point."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
w = init_encoder_weights(T=64, U=256, L=2, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=1)
ew = EncoderWeights(ctx, 64, 256, 2, True, -3.0).set_from_arrays(w)
x = torch.rand((n, 64), device="cuda") * 0.5 + 0.2
q = torch.empty((n, 5), device="cuda"); ls = torch.empty((n, 64), device="cuda")
lib = _lib.load()
def run():
    _lib.check(lib.qbold_encoder_fused_fwd(ctx.handle, C.byref(ew.shape), ew.fused_ptr(), _ptr(x), _ptr(q), _ptr(ls), n, _stream()), "fused")
for _ in range(30): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
print(f"fused encoder: {a.elapsed_time(b) / 20:.4f} ms per {n} voxels")
