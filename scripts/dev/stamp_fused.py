"""Dev: per-op cycle stamps of wide_fused_kernel (needs a -DQB_FUSED_STAMP build of the library)."""
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
w = init_encoder_weights(T=64, U=256, L=2, channelwise_gating=True, resid_init_std=0.05, im_loss_sigma=0.05, seed=1)
ew = EncoderWeights(ctx, 64, 256, 2, True, -3.0).set_from_arrays(w)
x = torch.rand((n, 64), device="cuda") * 0.5 + 0.2
q = torch.empty((n, 5), device="cuda"); ls = torch.empty((n, 64), device="cuda")
lib = _lib.load()
def run():
    _lib.check(lib.qbold_encoder_fused_fwd(ctx.handle, C.byref(ew.shape), ew.fused_ptr(), _ptr(x), _ptr(q), _ptr(ls), n, _stream()), "fused")
for _ in range(20): run()
torch.cuda.synchronize()
st = torch.zeros(8192, dtype=torch.int64, device="cuda")
os.environ["QBOLD_FUSED_STAMPS"] = str(st.data_ptr())
run(); torch.cuda.synchronize()
del os.environ["QBOLD_FUSED_STAMPS"]
s = st.cpu().numpy(); s = s[s != 0]
per = 12  # stamps per pass: start, converted, first layer, 4 per block x 2, head
names = ["convert", "first", "skip0", "t0", "r0", "gate0", "skip1", "t1", "r1", "gate1", "head"]
s = s[: (len(s) // per) * per].reshape(-1, per)
d = np.diff(s, axis=1)
print("passes:", len(s), " pass-to-pass cycles (median):", int(np.median(np.diff(s[:, 0]))))
for k, nm in enumerate(names):
    print(f"{nm:8s} median {int(np.median(d[:, k])):7d}  min {int(d[:, k].min()):7d}  max {int(d[:, k].max()):7d}")
print("gap head-end -> next pass start:", int(np.median(s[1:, 0] - s[:-1, -1])))
