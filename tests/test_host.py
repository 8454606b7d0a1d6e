"""CPU tests of the host side: configuration schema against the fixture produced by the
reference's own utils.load_arguments, the C-ABI export list against include/qbold_hip.h, the
host-only context (tau grid, F(x) table) and loud failure without a GPU."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", autouse=True)
def built():
    from qbold_vi_amd.build import build_lib
    build_lib()


def test_merged_config_matches_reference_fixture():
    from qbold_vi_amd.utils import load_arguments
    gold = json.load(open(os.path.join(GOLD, "merged_config_optimal.json")))
    args = load_arguments(["qbold_train_model.py", os.path.join(ROOT, "configurations", "optimal.yaml")])
    assert len(gold) == 37
    for k, v in gold.items():
        assert args[k] == v["value"], k
        assert type(args[k]).__name__ == v["type"], k
    # PyYAML hands over '2e-3' as a string; the float comes from type(default)(value) (Appendix B6)
    assert args["pt_lr"] == 0.002 and isinstance(args["pt_lr"], float)
    assert args["student_t_df"] == 200 and args["gate_offset"] == -3.0 and args["name"] == "optimal"
    assert args["use_population_prior"] is False  # falsy default -> raw YAML value


def test_defaults_and_cli_flags():
    from qbold_vi_amd.utils import get_defaults, load_arguments, setup_argparser
    d = get_defaults()
    assert d["use_population_prior"] is True and d["wandb_project"] == ""
    t = get_defaults("train")
    assert t["use_population_prior"] is False and t["use_wandb"] is True and "wandb_project" not in t
    args = load_arguments(["train.py", "--no_units", "12", "--pt_lr", "0.01", "-d", "/tmp/x"], entry="train")
    assert args["no_units"] == 12 and args["pt_lr"] == 0.01 and args["d"] == "/tmp/x"
    assert args["synthetic_voxels"] == 0 and args["mc_samples"] == 1 and args["devices"] == 1
    # `--flag False` is truthy exactly as in the reference (type=bool; Appendix B6)
    a = vars(setup_argparser(d).parse_args(["--use_mvg", "False"]))
    assert a["use_mvg"] is True


def test_ini_config_has_reference_keys(params):
    for k in ("tr", "ti", "te", "tau_start", "tau_end", "tau_step", "dchi", "gamma", "b0", "t1b", "r2t",
              "td", "nb", "hct", "s0", "simulate_noise", "tau_weighted", "snr", "oef_start", "oef_end",
              "oef_mean", "oef_std", "dbv_start", "dbv_end", "dbv_mean", "dbv_std", "sample_size"):
        assert k in params
    assert float(params["gamma"]) == 2.67513e8 and params["simulate_noise"] == "True"


def test_library_exports_every_declared_symbol():
    from qbold_vi_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "qbold_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(qbold_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/qbold_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().qbold_abi_version() == 5      # v5: qbold_encoder_shape.layer_norm, dropout_rate, dropout_seed


def test_host_only_context_tau_grid_and_table(params, oracle32, oracle64):
    from qbold_vi_amd.ops import Context
    ctx = Context(params, host_only=True)
    assert ctx.T == 11 and ctx.se_idx == 2
    np.testing.assert_array_equal(ctx.taus, oracle32.taus)
    x = np.linspace(0, 0.064 * 301.74327 * 0.999, 4001).astype(np.float32)
    F, dF = ctx.table_eval(x)
    # the table reproduces the reference's float32 Simpson sum to its own rounding noise (~1e-5)
    assert np.abs(F - oracle32.tissue_F(x)).max() < 4e-5
    # ... and is a 1e-6 approximation of the float64 sum with node 0 removed (Appendix B3)
    u0 = 1e-5
    node0 = (2 + u0) * np.sqrt(1 - u0) * (1 - np.cos(0) + (1.5 * x.astype(np.float64) * u0) ** 2 / 4) \
        / (3 * u0 * u0) * ((1 - 1e-5) / 128) / 3
    sem = oracle64.tissue_F(x) - node0
    assert np.abs(F - sem).max() < 3e-6
    # derivative = J1-kernel sum over all nodes (TensorFlow's gradient of bessel_j0)
    xs = x[::100]
    assert np.abs(dF[::100] - oracle64.tissue_dF(xs)).max() < 2e-5
    with pytest.raises(Exception):
        ctx.signal_fwd(np.zeros((2, 2), np.float32))  # not even a tensor


def test_other_tau_grids(params):
    from qbold_vi_amd.ops import Context
    p = dict(params, tau_start="-0.015", tau_end="0.065", tau_step="0.00125")  # T=64
    ctx = Context(p, host_only=True)
    assert ctx.T == 64 and ctx.se_idx == 12 and abs(ctx.taus[12]) < 1e-9
    p24 = dict(params, tau_start="-0.028", tau_end="0.065", tau_step="0.004")
    assert Context(p24, host_only=True).T == 24
    from qbold_vi_amd._lib import QboldError
    with pytest.raises(QboldError):
        Context(dict(params, tau_step="0.0001"), host_only=True)  # > 64 taus


def test_fails_loudly_without_gpu(params):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from qbold_vi_amd import EncoderTrainer, SignalGenerationLayer
    from qbold_vi_amd._lib import QboldError
    from qbold_vi_amd.ops import Context
    with pytest.raises(QboldError):
        Context(params)
    with pytest.raises(QboldError):
        SignalGenerationLayer(params, True, True)
    with pytest.raises(QboldError):
        EncoderTrainer(params, activation_type='relu', use_population_prior=False)
    # a host-only context cannot launch anything
    ctx = Context(params, host_only=True)
    with pytest.raises(QboldError):
        ctx.signal_fwd(torch.zeros((4, 2)))
    from qbold_vi_amd import _lib
    rc = _lib.load().qbold_signal_fwd(ctx.handle, None, None, 4, None)
    assert rc == -4  # QBOLD_ERR_NO_DEVICE


def test_encoder_param_count_matches_survey():
    from qbold_vi_amd import _lib
    lib = _lib.load()
    s = _lib.EncoderShape(11, 60, 2, 1, -3.0, 1)
    assert lib.qbold_encoder_num_params(C.byref(s)) == 30976   # centre-tap subset
    s1 = _lib.EncoderShape(11, 60, 2, 0, 0.0, 1)
    assert lib.qbold_encoder_num_params(C.byref(s1)) == 30976 - 2 * (60 * 60 + 60) + 2 * (60 + 1)
    assert lib.qbold_encoder_packed_floats(C.byref(s)) * 4 < 150 * 1024  # fits the 160 KiB LDS
    s9 = _lib.EncoderShape(11, 60, 2, 1, -3.0, 9)   # full 3x3x1 residual kernels
    assert lib.qbold_encoder_num_params(C.byref(s9)) == 146176  # SURVEY 2.1 parameter count


def test_shard_ranges():
    from qbold_vi_amd.distributed import shard_range
    for n in (0, 1, 7, 1000, 1 << 20):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_c_program_links_against_the_abi(tmp_path):
    """include/qbold_hip.h is plain C: compile tests/c/abi_smoke.c with gcc and run it."""
    import shutil
    import subprocess
    from qbold_vi_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = tmp_path / "abi_smoke"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe),
                           "-L", libdir, "-lqbold_hip", "-lm", f"-Wl,-rpath,{libdir}",
                           "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi ok" in r.stdout


def test_nifti_writer_follows_the_nifti1_layout(tmp_path):
    """The export format of save_predictions (model.py:790-801) without nibabel: field offsets per
    the NIfTI-1.1 header definition, Fortran-ordered voxels, gzip container, round trip."""
    import gzip
    import struct
    from qbold_vi_amd import nifti
    rng = np.random.default_rng(0)
    arr = rng.normal(size=(5, 4, 3, 2)).astype(np.float32)
    path = tmp_path / "a.nii.gz"
    nifti.save(arr, str(path))
    raw = gzip.open(path, "rb").read()
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:348] == b"n+1\0"
    assert struct.unpack_from("<8h", raw, 40) == (4, 5, 4, 3, 2, 1, 1, 1)          # dim
    assert struct.unpack_from("<hh", raw, 70) == (16, 32)                            # FLOAT32, bitpix
    assert struct.unpack_from("<f", raw, 108)[0] == 352.0                            # vox_offset
    assert struct.unpack_from("<hh", raw, 252) == (0, 0)                             # qform/sform codes
    assert len(raw) == 352 + arr.size * 4
    # x varies fastest on disk
    vox = np.frombuffer(raw, "<f4", arr.size, 352)
    assert vox[1] == arr[1, 0, 0, 0] and vox[5] == arr[0, 1, 0, 0]
    back, hdr = nifti.load(str(path))
    np.testing.assert_array_equal(back, arr)
    # a template header donates orientation; dims / datatype are rewritten
    tmpl = bytearray(hdr)
    struct.pack_into("<hh", tmpl, 252, 1, 1)
    struct.pack_into("<8f", tmpl, 76, 1.0, 2.0, 2.0, 3.0, 1.0, 1.0, 1.0, 1.0)
    # ... but not its intensity scaling: an example image that is a scaled int16 scan (scl_slope 0.25, offset 7,
    # a display window) must not rescale the float maps exported under its header
    struct.pack_into("<ff", tmpl, 112, 0.25, 7.0)
    struct.pack_into("<ff", tmpl, 124, 900.0, 100.0)
    struct.pack_into("<h", tmpl, 68, 1005)
    nifti.save(arr[..., 0].astype(np.float64), str(tmp_path / "b.nii"), bytes(tmpl))
    b, hb = nifti.load(str(tmp_path / "b.nii"))
    assert b.dtype == np.float64 and b.shape == (5, 4, 3)
    np.testing.assert_array_equal(b, arr[..., 0].astype(np.float64))
    assert struct.unpack_from("<hh", hb, 252) == (1, 1) and struct.unpack_from("<8f", hb, 76)[1:4] == (2.0, 2.0, 3.0)
    assert struct.unpack_from("<ff", hb, 112) == (1.0, 0.0) and struct.unpack_from("<ff", hb, 124) == (0.0, 0.0)
    assert struct.unpack_from("<h", hb, 68)[0] == 0
    with pytest.raises(ValueError):
        nifti.save(arr.astype(np.complex64), str(tmp_path / "c.nii"))


class _FakeH5Node(dict):
    """Stand-in for h5py groups / files: a mapping with .attrs; 'a/b' paths resolve like h5py's."""

    def __init__(self):
        super().__init__()
        self.attrs = {}

    def __getitem__(self, key):
        node = self
        for part in key.split("/"):
            node = dict.__getitem__(node, part)
        return node


def test_keras_h5_mapping_by_order_and_shape():
    """keras_h5.py against an in-memory stand-in for the h5py objects (h5py itself is absent here):
    Keras layout -> canonical names -> Keras layout, including layer names that do not start at
    conv3d_1 (Keras numbers layers per process) and the 4-output head of use_mvg=False."""
    from qbold_vi_amd import keras_h5
    from qbold_vi_amd.init import init_encoder_weights
    w = init_encoder_weights(T=11, U=12, L=2, channelwise_gating=True, seed=4, spatial_taps=9)
    rng = np.random.default_rng(0)
    for k in ("b0", "bc", "br1", "br2", "bg", "bf"):
        w[k] = rng.normal(size=w[k].shape).astype(np.float32)
    var = keras_h5.canonical_to_variables(w)
    assert len(var) == 6 + 8 * 2 and var[0][1].shape == (1, 1, 1, 11, 12) and var[4][1].shape == (3, 3, 1, 12, 12)
    # build the file as Keras would after an earlier model shifted the numbering by 7
    f = _FakeH5Node()
    shift = lambda n: n if n.startswith("conv3d/") else "conv3d_%d/%s" % (int(n.split("/")[0].split("_")[1]) + 7, n.split("/")[1])
    groups = [("input_1", []), ("lambda", []), ("conv3d", var[:2]), ("model", var[2:-2]), ("conv3d_sigma", var[-2:])]
    f.attrs["layer_names"] = [g.encode() for g, _ in groups]
    for gname, items in groups:
        g = _FakeH5Node()
        g.attrs["weight_names"] = [shift(n).encode() for n, _ in items]
        for n, a in items:
            node = g
            parts = shift(n).split("/")
            for part in parts[:-1]:
                node = node.setdefault(part, _FakeH5Node()) if part not in node else dict.__getitem__(node, part)
            dict.__setitem__(node, parts[-1], a)
        dict.__setitem__(f, gname, g)
    back = keras_h5.variables_to_canonical(keras_h5.flatten_variables(f))
    for k in w:
        if k != "meta":
            np.testing.assert_array_equal(back[k], w[k])
    # Keras writes a Functional model's variables by graph depth, not by creation: inside a block the 3x3x1
    # kernels come before the shared 1x1x1 conv and the gate.  With numbered layer names the creation numbers
    # restore the order; with unnumbered names the kernel shapes do.
    inner = var[2:-2]
    depth_order = []
    for l in range(2):
        b = inner[8 * l: 8 * l + 8]                      # Wc bc Wr1 br1 Wr2 br2 Wg bg
        depth_order += b[2:6] + b[0:2] + b[6:8]
    depth_order += inner[16:]
    for rename in (shift, lambda n: "layer_%s/%s" % (n.split("/")[0].replace("conv3d", "x"), n.split("/")[1])):
        f2 = _FakeH5Node()
        groups2 = [("conv3d", var[:2]), ("model", depth_order), ("sigma", var[-2:])]
        f2.attrs["layer_names"] = [g.encode() for g, _ in groups2]
        for gname, items in groups2:
            g = _FakeH5Node()
            g.attrs["weight_names"] = [rename(n).encode() for n, _ in items]
            for n, a in items:
                node = g
                parts = rename(n).split("/")
                for part in parts[:-1]:
                    node = node.setdefault(part, _FakeH5Node()) if part not in node else dict.__getitem__(node, part)
                dict.__setitem__(node, parts[-1], a)
            dict.__setitem__(f2, gname, g)
        back2 = keras_h5.variables_to_canonical(keras_h5.flatten_variables(f2))
        for k in w:
            if k != "meta":
                np.testing.assert_array_equal(back2[k], w[k], err_msg=k)
    # diagonal family: 4-column final layer
    w4 = dict(w, Wf=w["Wf"][:, :4], bf=w["bf"][:4])
    assert keras_h5.variables_to_canonical([a for _, a in keras_h5.canonical_to_variables(w4)])["Wf"].shape == (12, 4)
    # wrong variable count / shapes are rejected
    with pytest.raises(ValueError):
        keras_h5.variables_to_canonical([a for _, a in var[:-1]])
    bad = [a for _, a in var]
    bad[4] = bad[4][:, :, :, :, :5]
    with pytest.raises(ValueError):
        keras_h5.variables_to_canonical(bad)
    # without h5py the file entry points say what is missing
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            keras_h5.load_keras_h5("nope.h5")


def test_crop_dataset_windows_match_the_volumes():
    """train.prepare_dataset (train.py:17-72) as CropDataset: blank crop [17:-17, 10:-10], masking,
    random crop_size x crop_size windows over (X, Y) that keep every slice and channel, batches of
    38 (training) / 3 (validation, subjects in order).  Host index arithmetic only: runs on CPU
    tensors with a stand-in model."""
    torch = pytest.importorskip("torch")
    from qbold_vi_amd.training import CropDataset

    class Model:
        def predict(self, x, want=("out1",)):
            # a "prior" that encodes where each voxel came from: (x index, y index, subject, 0, 0)
            return [self.tag.to(x.dtype)]

    rng = np.random.default_rng(0)
    S, X, Y, Z, T = 3, 60, 44, 8, 11
    vol = rng.uniform(0.5, 1.5, (S, X, Y, Z, T + 1)).astype(np.float32)
    vol[..., -1] = (rng.uniform(size=(S, X, Y, Z)) > 0.3)
    m = Model()
    xs, ys, ss = np.meshgrid(np.arange(X - 34), np.arange(Y - 20), np.arange(S), indexing="ij")
    tag = np.zeros((S, X - 34, Y - 20, Z, 5), np.float32)
    tag[..., 0] = xs.transpose(2, 0, 1)[..., None]
    tag[..., 1] = ys.transpose(2, 0, 1)[..., None]
    tag[..., 2] = ss.transpose(2, 0, 1)[..., None]
    m.tag = torch.as_tensor(tag)
    ds = CropDataset(torch.as_tensor(vol), m, crop_size=10, training=True)
    assert ds.data.shape == (S, X - 34, Y - 20, Z, T) and ds.batch == 38
    g = torch.Generator().manual_seed(1)
    x5, m5, p5 = ds.next_batch(g)
    assert x5.shape == (38, 10, 10, Z, T) and m5.shape == (38, 10, 10, Z) and p5.shape == (38, 10, 10, Z, 5)
    inner = vol[:, 17:-17, 10:-10]
    for b in range(38):
        x0, y0, s = int(p5[b, 0, 0, 0, 0]), int(p5[b, 0, 0, 0, 1]), int(p5[b, 0, 0, 0, 2])
        win = inner[s, x0:x0 + 10, y0:y0 + 10]
        np.testing.assert_array_equal(m5[b].numpy(), win[..., -1])
        np.testing.assert_array_equal(x5[b].numpy(), win[..., :-1] * win[..., -1:])
        # the prior travels with its voxels
        np.testing.assert_array_equal(p5[b, :, :, 0, 0].numpy(), (x0 + np.arange(10))[:, None] * np.ones((1, 10)))
    # crops larger than the volume are clamped (train.py:24); validation walks the subjects in order
    m.tag = torch.zeros((S, X, Y, Z, 5))
    dv = CropDataset(torch.as_tensor(vol), m, crop_size=76, training=False, blank_crop=False)
    assert dv.crop == [60, 44] and dv.batch == 3
    xa, ma, _ = dv.next_batch(g)
    np.testing.assert_array_equal(ma.numpy(), vol[..., -1])


def test_hand_placed_lds_reads_are_not_touched_before_their_wait(tmp_path):
    """The weight-streaming encoder kernels read their LDS rings through inline-asm ds_read + counted s_waitcnt
    (the compiler would otherwise drain the LDS-direct loads in flight).  The compiler does not know those reads
    are asynchronous: scripts/check_lds_hazards.py scans the ISA for any use of a register between the read that
    writes it and the wait that covers it (a bias read into one vector component once made the compiler copy the
    pending register: garbage biases, run to run different)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "qbold_vi_amd", "csrc")
    # (wide_kernels.hip's wide_dense_kernel passes the same check -- 592 reads, 0 hazards -- but takes minutes to
    # compile; QB_FUSED_DEV builds the config-3 instantiation of the one-launch kernel only)
    for src, flags, key in (("wide_fused_kernels.hip", ["-DQB_FUSED_DEV"], "wide_fused_kernel"),):
        out = tmp_path / (src + ".s")
        subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-gpu-rdc", "-Wno-unused-function",
                               "--cuda-device-only", "-S", *flags, os.path.join(csrc, src), "-o", str(out)],
                              stderr=subprocess.DEVNULL)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_lds_hazards.py"), str(out), key],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        assert "LDS reads checked, 0 hazards" in r.stdout and " 0 LDS reads" not in r.stdout, r.stdout
        # the weight ring's handshakes wait with per-site vmcnt allowances (signal loads and head stores known to
        # be younger than the awaited stage, wide_fused_kernels.hip sync_extras): replayed against the ISA; an
        # allowance one too large at any site must be caught
        chk = os.path.join(ROOT, "scripts", "check_vmcnt_ring.py")
        r = subprocess.run([sys.executable, chk, str(out), key], capture_output=True, text=True)
        assert r.returncode == 0 and " 0 violations" in r.stdout, r.stdout[-2000:]
        assert int(r.stdout.split(" handshakes replayed")[0].split()[-1]) > 400
        text = out.read_text()
        import re
        sites = sorted({int(n) for n in re.findall(r"s_waitcnt vmcnt\((\d+)\)\n\s*;;#ASMEND\n\s*s_barrier", text)})
        assert len(sites) > 3, sites          # several different allowances are in use
        for n in sites:
            bad = tmp_path / "bad.s"
            bad.write_text(re.sub(r"s_waitcnt vmcnt\(%d\)(\n\s*;;#ASMEND\n\s*s_barrier)" % n,
                                  "s_waitcnt vmcnt(%d)\\1" % (n + 1), text))
            r = subprocess.run([sys.executable, chk, str(bad), key], capture_output=True, text=True)
            assert r.returncode == 1 and "still has" in r.stdout, (n, r.stdout[-500:])


def test_build_verified_the_isa_of_every_ring_instantiation():
    """qbold_vi_amd/build.py compiles wide_fused_kernels.hip with -save-temps, runs both static checkers on the ISA
    of EVERY instantiation and refuses to keep an object that fails: the verdict of the very object linked into
    libqbold_hip.so is the stamp beside it (both V4 instantiations replayed -- <4, 1, true> too --, no hazards in any
    of the six kernels)."""
    from qbold_vi_amd import build
    obj = os.path.join(build.OBJ, "wide_fused_kernels.o")
    stamp = obj + ".isa_ok"
    if not os.path.exists(obj):
        pytest.skip("library was not built in this tree")
    assert os.path.exists(stamp) and os.path.getmtime(stamp) >= os.path.getmtime(obj)
    text = open(stamp).read()
    for key in build.ISA_CHECKS["wide_fused_kernels.hip"]["ring"]:
        line = next(l for l in text.splitlines() if "check_vmcnt_ring.py " + key in l)
        assert " 0 violations" in line and int(line.split(" handshakes replayed")[0].split()[-1]) > 200, line
    for key in build.ISA_CHECKS["wide_fused_kernels.hip"]["hazard"]:
        line = next(l for l in text.splitlines() if "check_lds_hazards.py " + key in l)
        assert " 0 hazards" in line and int(line.split(": ")[1].split()[0]) > 1000, line
    assert {k[-12:] for k in build.ISA_CHECKS["wide_fused_kernels.hip"]["ring"]} == {"ILi4ELi1ELb1", "ILi4ELi2ELb1"}


def test_training_backward_says_when_it_recomputes(params):
    """qbold_encoder_train_bwd_recomputes is the one predicate both sides of the training step use: the one-launch
    forward leaves out exactly what the backward says it will recompute (2: skip, gate logits, t and r)."""
    from qbold_vi_amd import _lib
    from qbold_vi_amd.ops import Context
    lib = _lib.load()
    ctx = Context(params, host_only=True)
    yes = _lib.EncoderShape(11, 60, 2, 1, -3.0, 9)
    assert lib.qbold_encoder_train_bwd_recomputes(ctx.handle, C.byref(yes), 1 << 20) == 2
    assert lib.qbold_encoder_train_bwd_recomputes(ctx.handle, C.byref(yes), 1 << 23) == 0      # 32-bit row offsets
    assert lib.qbold_encoder_train_bwd_recomputes(ctx.handle, C.byref(yes), 0) == 0
    for shape in (_lib.EncoderShape(11, 256, 2, 1, -3.0, 1),      # wide: layer-wise kernels
                  _lib.EncoderShape(11, 60, 3, 1, -3.0, 1),       # three blocks: not an LDS-resident shape
                  _lib.EncoderShape(11, 60, 2, 0, -3.0, 1),       # shared gate
                  _lib.EncoderShape(24, 60, 2, 1, -3.0, 1),       # another protocol than the context's
                  _lib.EncoderShape(11, 60, 2, 1, -3.0, 1, 1)):   # bf16 encoder mode
        assert lib.qbold_encoder_train_bwd_recomputes(ctx.handle, C.byref(shape), 1 << 20) == 0
    assert lib.qbold_encoder_train_bwd_recomputes(None, C.byref(yes), 1 << 20) == 0


def test_keras_h5_carries_the_inverse_gamma_hyper_prior():
    """A checkpoint saved with infer_inv_gamma=True holds the tfp VariableLayer's one rank-1 variable of four logs
    (model.py:201-205) in a group of its own: mapped to 'hyper_prior' and back."""
    from qbold_vi_amd import keras_h5
    from qbold_vi_amd.init import init_encoder_weights
    w = init_encoder_weights(T=11, U=12, L=1, channelwise_gating=True, seed=4, spatial_taps=9)
    w["Wf"], w["bf"] = w["Wf"][:, :4].copy(), w["bf"][:4].copy()          # the diagonal family it runs with
    w["hyper_prior"] = np.log(np.array([20.0, 2.5, 15.0, 3.0], np.float32))
    var = keras_h5.canonical_to_variables(w)
    assert var[-1][0] == "variable_layer/constant:0" and len(var) == 6 + 8 + 1
    f = _FakeH5Node()
    groups = [("conv3d", var[:2]), ("model", var[2:-3] + var[-1:]), ("conv3d_6", var[-3:-1])]
    f.attrs["layer_names"] = [g.encode() for g, _ in groups]
    for gname, items in groups:
        g = _FakeH5Node()
        g.attrs["weight_names"] = [n.encode() for n, _ in items]
        for n, a in items:
            node, parts = g, n.split("/")
            for part in parts[:-1]:
                node = node.setdefault(part, _FakeH5Node()) if part not in node else dict.__getitem__(node, part)
            dict.__setitem__(node, parts[-1], a)
        dict.__setitem__(f, gname, g)
    back = keras_h5.variables_to_canonical(keras_h5.flatten_variables(f))
    np.testing.assert_array_equal(back["hyper_prior"], w["hyper_prior"])
    assert back["Wf"].shape == (12, 4)
    for k in ("W0", "Wc", "Wr1", "Wg", "Ws", "bs"):
        np.testing.assert_array_equal(back[k], w[k])
    # without the variable nothing changes
    w.pop("hyper_prior")
    assert "hyper_prior" not in keras_h5.variables_to_canonical([a for _, a in keras_h5.canonical_to_variables(w)])


def test_digamma_matches_scipy():
    from scipy.special import digamma as ref
    from qbold_vi_amd.training import digamma
    for x in (1e-3, 0.1, 0.5, 1.0, 2.5, 5.999, 6.0, 20.0, 1e3, 1e6):
        assert abs(digamma(x) - float(ref(x))) < 1e-12 * max(1.0, abs(float(ref(x)))), x
    with pytest.raises(ValueError):
        digamma(0.0)
