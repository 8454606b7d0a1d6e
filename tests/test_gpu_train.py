"""End-to-end training on the GPU through the reference-shaped entry points: pre-training on
synthetic voxels, ELBO fine-tuning, checkpoints and phase skipping."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_config(tmp, **over):
    from qbold_vi_amd.utils import load_arguments
    args = load_arguments(["train.py", os.path.join(ROOT, "configurations", "optimal.yaml")], entry="train")
    args.update(no_units=24, no_intermediate_layers=1, no_pt_epochs=80, no_ft_epochs=3,
                save_directory=str(tmp), synthetic_voxels=20000, mc_samples=2)
    args.update(over)
    return args


def test_lr_schedule_is_the_reference_linear_ramp():
    from qbold_vi_amd.training import lr_schedule
    assert lr_schedule(5e-3, 0) == 5e-3
    assert abs(lr_schedule(5e-3, 2000) - (5e-3 + (5e-5 - 5e-3) * 0.5)) < 1e-12
    assert abs(lr_schedule(5e-3, 4000) - 5e-5) < 1e-12
    assert abs(lr_schedule(2e-4, 1) - (2e-4 + (2e-6 - 2e-4) / 4000)) < 1e-15


def test_two_phase_training_and_phase_skipping(tmp_path, monkeypatch):
    from qbold_vi_amd import training
    monkeypatch.chdir(ROOT)   # the INI `config` is read from the CWD, as in the reference
    cfg = small_config(tmp_path)
    model, trainer, hist = training.train_model(cfg, pt_sample_size=200)
    pt = [h for h in hist if "val_oef_metric" in h]
    ft = [h for h in hist if "val_elbo" in h]
    assert len(pt) == 80 and len(ft) == 3
    # supervised phase learns (one 40,000-voxel step per epoch here): the loss falls steadily and
    # the OEF / DBV errors of the posterior mean shrink
    assert pt[-1]["loss"] < pt[0]["loss"] - 10.0 and pt[-1]["val_loss"] < pt[0]["val_loss"] - 10.0
    assert pt[-1]["val_oef_metric"] < 0.9 * pt[0]["val_oef_metric"]
    assert pt[-1]["val_dbv_metric"] < 0.75 * pt[0]["val_dbv_metric"]
    # ELBO phase: objective falls and every reference metric key is present (train.py:350-355)
    assert ft[-1]["loss"] < ft[0]["loss"]
    for k in ("val_nll", "val_elbo", "val_elbo_smooth", "val_smoothness", "val_smoothness_scaled", "val_kl"):
        assert k in ft[-1]
    assert abs(ft[-1]["val_elbo"] - (ft[-1]["val_nll"] + ft[-1]["val_kl"])) < 1e-9
    assert os.path.isfile(tmp_path / "pt_model.npz") and os.path.isfile(tmp_path / "final_model.npz")
    lines = [json.loads(l) for l in open(tmp_path / "metrics.jsonl")]
    assert len(lines) == 83
    # second call: both phases are skipped because their weight files exist (train.py:197-199,260)
    w_before = {k: v.copy() for k, v in model.get_weights().items()}
    model2, _, hist2 = training.train_model(cfg)
    assert hist2 == []
    for k, v in model2.get_weights().items():
        np.testing.assert_array_equal(v, w_before[k])
    # ModelBuilder reports the three-state status (qbold_build_model.py:45-56)
    mb = training.ModelBuilder(dict(cfg, save_directory=os.path.relpath(tmp_path, ROOT)))
    assert mb.weight_status == training.WeightStatus.FULL_TRAINED
    os.remove(tmp_path / "final_model.npz")
    mb = training.ModelBuilder(dict(cfg, save_directory=os.path.relpath(tmp_path, ROOT)))
    assert mb.weight_status == training.WeightStatus.PRE_TRAINED


def test_pretraining_leaves_the_stream2_only_tensors_untouched(tmp_path, monkeypatch):
    """Keras' loss=[synthetic_data_loss, None, None] (train.py:388-392) gives the residual / gating convolutions
    and the sigma head a None gradient: apply_gradients skips them, so tfa's AdamW neither updates nor DECAYS
    them.  After pre-training with pt_adamw_decay > 0 they are bit-identical to their initial values, while
    every stream-1 tensor has moved."""
    from qbold_vi_amd import training
    monkeypatch.chdir(ROOT)
    cfg = small_config(tmp_path, no_pt_epochs=3, no_ft_epochs=0, pt_adamw_decay=2e-2, use_swa=True)
    params = training.get_params()
    model0, _, _ = training.create_encoder_model(cfg, params)   # the initialisers are seeded: same start below
    w0 = {k: v.copy() for k, v in model0.get_weights().items()}
    model, _, _ = training.create_and_train_on_synthetic_data(cfg, params, log=training.MetricsLog(echo=False),
                                                              sample_size=200)
    w1 = model.get_weights()
    for k in ("Wr1", "br1", "Wr2", "br2", "Wg", "bg", "Ws", "bs"):
        np.testing.assert_array_equal(w1[k], w0[k], err_msg=k)
    for k in ("W0", "Wc", "Wf"):
        assert np.abs(w1[k] - w0[k]).max() > 1e-4, k


def test_sweep_style_configuration_trains(tmp_path, monkeypatch):
    """The hyper-parameters the reference's sweep file fixes (configurations/sweep_prior.yaml):
    Student-t likelihood with df = 2, three-image normalisation, an inverse-gamma prior on the
    pre-training variances, 30 units -- every switch off the optimal.yaml path, through both phases."""
    from qbold_vi_amd import training
    monkeypatch.chdir(ROOT)
    cfg = small_config(tmp_path, no_units=30, student_t_df=2, multi_image_normalisation=True,
                       inv_gamma_alpha=3.0, inv_gamma_beta=0.15, no_pt_epochs=40, no_ft_epochs=3)
    model, trainer, hist = training.train_model(cfg, pt_sample_size=200)
    pt = [h for h in hist if "val_oef_metric" in h]
    ft = [h for h in hist if "val_elbo" in h]
    assert len(pt) == 40 and len(ft) == 3
    assert all(np.isfinite(h["loss"]) for h in hist)
    assert pt[-1]["loss"] < pt[0]["loss"] - 5.0
    assert ft[-1]["loss"] < ft[0]["loss"]
    assert trainer.context.lib is not None and trainer._student_t_df == 2


def test_r2p_loss_pretraining(tmp_path, monkeypatch):
    """use_r2p_loss=True (train.py:125, 388; model.py:475-490): the pre-training loss carries the R2' term
    and its gradient; the R2' error of the stream-1 predictions falls along with the loss."""
    from qbold_vi_amd import training
    monkeypatch.chdir(ROOT)
    cfg = small_config(tmp_path, use_r2p_loss=True, no_pt_epochs=40, no_ft_epochs=1)
    model, trainer, hist = training.train_model(cfg, pt_sample_size=200)
    pt = [h for h in hist if "val_oef_metric" in h]
    assert len(pt) == 40 and all(np.isfinite(h["loss"]) for h in hist)
    assert pt[-1]["loss"] < pt[0]["loss"] - 3.0
    assert pt[-1]["val_r2p_metric"] < 0.75 * pt[0]["val_r2p_metric"]


def test_reference_argparse_defaults_train(tmp_path, monkeypatch):
    """`python train.py` without a YAML uses get_defaults() (train.py:150-187): the diagonal family
    (use_mvg=False), Student-t df = 2, log data, 3-image normalisation, 30 units, one block."""
    from qbold_vi_amd import training
    from qbold_vi_amd.utils import load_arguments
    monkeypatch.chdir(ROOT)
    args = load_arguments(["train.py"], entry="train")
    assert args["use_mvg"] is False and args["predict_log_data"] is True and args["student_t_df"] == 2
    args.update(save_directory=str(tmp_path), synthetic_voxels=20000, no_pt_epochs=40, no_ft_epochs=3, pt_lr=2e-3)
    model, trainer, hist = training.train_model(args, pt_sample_size=200)
    pt = [h for h in hist if "val_oef_metric" in h]
    ft = [h for h in hist if "val_elbo" in h]
    assert len(pt) == 40 and len(ft) == 3 and all(np.isfinite(h["loss"]) for h in hist)
    assert pt[-1]["loss"] < pt[0]["loss"] - 3.0
    assert ft[-1]["loss"] < ft[0]["loss"]
    w = model.get_weights()
    assert w["Wf"].shape == (30, 4)
    assert np.abs(model.weights.to_arrays()["Wf"][:, 4]).max() == 0.0    # the unused head column never moves


def test_wide_encoder_trains(tmp_path, monkeypatch):
    """no_units beyond the LDS-resident kernels (128: weight-streaming inference, layer-wise training
    GEMMs in 64 x 64 slabs) through both training phases."""
    from qbold_vi_amd import training
    monkeypatch.chdir(ROOT)
    cfg = small_config(tmp_path, no_units=128, no_intermediate_layers=1, no_pt_epochs=25, no_ft_epochs=2,
                       synthetic_voxels=8192)
    model, trainer, hist = training.train_model(cfg, pt_sample_size=120)
    pt = [h for h in hist if "val_oef_metric" in h]
    ft = [h for h in hist if "val_elbo" in h]
    assert len(pt) == 25 and len(ft) == 2 and all(np.isfinite(h["loss"]) for h in hist)
    assert pt[-1]["loss"] < pt[0]["loss"] - 3.0
    assert model.weights.wide


def test_missing_real_data_directory_raises(tmp_path, monkeypatch):
    from qbold_vi_amd import training
    monkeypatch.chdir(ROOT)
    cfg = small_config(tmp_path, synthetic_voxels=0, d=str(tmp_path / "nope"), no_pt_epochs=1)
    with pytest.raises(Exception, match="Real data directory not found"):
        training.train_model(cfg, pt_sample_size=100)


def test_fine_tuning_on_image_crops(tmp_path, monkeypatch):
    """.npy volumes shaped [subj, X, Y, 8, T+2] as the reference loads them (train.py:208-221):
    random-crop pipeline, stream 2 with 3x3x1 convolutions, NLL + KL + 5 x TV objective."""
    from qbold_vi_amd import training
    from qbold_vi_amd.signals import SignalGenerationLayer
    monkeypatch.chdir(ROOT)
    params = training.get_params("config")
    layer = SignalGenerationLayer(dict(params, simulate_noise='True'), True, True)
    rng = np.random.default_rng(0)
    d = tmp_path / "data"
    os.makedirs(d)
    nx = ny = 14
    gx, gy = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny), indexing="ij")
    for name in ("ASE_scan", "ASE_INF", "ASE_SUP", "hyperv_ase", "baseline_ase"):
        # smooth parameter maps: neighbouring voxels are similar, as in a brain
        oef = 0.3 + 0.2 * gx[None, :, :, None] + 0.05 * rng.standard_normal((2, 1, 1, 8))
        dbv = 0.02 + 0.03 * gy[None, :, :, None] + 0.0 * oef
        y = np.stack([np.broadcast_to(oef, (2, nx, ny, 8)), np.broadcast_to(dbv, (2, nx, ny, 8))], -1)
        sig = layer(torch.as_tensor(y.reshape(-1, 2), dtype=torch.float32, device="cuda")).cpu().numpy()
        vol = np.concatenate([sig * 100.0, np.ones((sig.shape[0], 2), np.float32)], -1)
        vol = vol.reshape(2, nx, ny, 8, 13)
        vol[:, 0, :, :, -2:] = 0.0   # a slab outside the masks
        np.save(d / f"{name}.npy", vol)
    cfg = small_config(tmp_path / "run", synthetic_voxels=0, d=str(d), no_pt_epochs=40, no_ft_epochs=2,
                       crop_size=10)
    model, trainer, hist = training.train_model(cfg, pt_sample_size=200, max_ft_steps=None)
    ft = [h for h in hist if "val_elbo" in h]
    assert len(ft) == 2 and all(np.isfinite(h["val_elbo"]) for h in ft)
    assert ft[-1]["loss"] < ft[0]["loss"]
    assert ft[-1]["val_smoothness"] > 0.0 and "predictions_smoothness_metric" in ft[-1]
    assert abs(ft[-1]["val_elbo_smooth"] - (ft[-1]["val_elbo"] + 5.0 * ft[-1]["val_smoothness"])) < 1e-9
    # NIfTI export set of save_predictions (model.py:772-887; train.py:248-251, 272-279)
    from qbold_vi_amd import nifti
    run = tmp_path / "run"
    for stem in ("pt_baseline", "pt_hyperv"):
        for suffix in ("_oef", "_dbv", "_r2p", "_logstds"):
            assert (run / f"{stem}{suffix}.nii.gz").is_file()
    for suffix, chans in (("_oef", 1), ("_dbv", 1), ("_r2p", 1), ("_logstds", 3), ("_likelihood", 1),
                          ("_kl", 1), ("_residual", 1)):
        im, _ = nifti.load(str(run / f"baseline{suffix}.nii.gz"))
        assert im.shape == (nx, ny, 8, 2 * chans) and im.dtype == np.float32 and np.isfinite(im).all()
    oef_map, _ = nifti.load(str(run / "baseline_oef.nii.gz"))
    assert 0.04 < oef_map.min() and oef_map.max() < 0.84
    kl_map, _ = nifti.load(str(run / "baseline_kl.nii.gz"))
    assert (kl_map[0] == 0).all() and (kl_map[1:] != 0).any()   # the slab outside the brain mask
    # whole-volume prediction through the spatial path
    vol = torch.as_tensor(np.load(d / "baseline_ase.npy")[..., :11], device="cuda")
    o1, o2, sg = model.predict(vol)
    assert o2.shape == (2, nx, ny, 8, 5) and sg.shape == (2, nx, ny, 8, 11) and bool(torch.isfinite(o2).all())


def test_train_py_cli(tmp_path):
    """`python train.py <yaml>` exactly as the reference is invoked (YAML values win over flags,
    train.py:473-480, so the save directory comes from the YAML)."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configurations", "optimal.yaml")))
    cfg.update(save_directory=str(tmp_path / "run"), no_pt_epochs=1, no_ft_epochs=1, no_units=16,
               no_intermediate_layers=1)
    ypath = tmp_path / "small.yaml"
    yaml.safe_dump(cfg, open(ypath, "w"))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), str(ypath),
                        "--synthetic_voxels", "4096"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.isfile(tmp_path / "run" / "pt_model.npz")
    assert os.path.isfile(tmp_path / "run" / "final_model.npz")
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert np.isfinite(last["val_elbo"])


def test_two_rank_training_matches_single_process(tmp_path):
    """train.py under torchrun with two ranks (sharing this box's one card over gloo -- the launch line
    the driver uses, with RCCL, on a real node): voxel shards per rank, all-reduced sums and gradients.
    Every rank applies the same update, so the result equals the single-process run up to the
    summation order of the gradient all-reduce."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configurations", "optimal.yaml")))
    cfg.update(no_pt_epochs=2, no_ft_epochs=2, no_units=16, no_intermediate_layers=1)
    env = dict(os.environ, PYTHONPATH=ROOT, QBOLD_DIST_BACKEND="gloo")
    runs = {}
    for name, launcher in (("one", [sys.executable]),
                           ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                    "--master-addr", "127.0.0.1", "--master-port", "29517"])):
        c = dict(cfg, save_directory=str(tmp_path / name))
        ypath = tmp_path / f"{name}.yaml"
        yaml.safe_dump(c, open(ypath, "w"))
        r = subprocess.run(launcher + [os.path.join(ROOT, "train.py"), str(ypath), "--synthetic_voxels", "4096"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        runs[name] = np.load(tmp_path / name / "final_model.npz")
    for k in runs["one"].files:
        a, b = runs["one"][k], runs["two"][k]
        assert np.isfinite(b).all()
        # AdamW divides by the root of the second moment: where a gradient component is near zero, the summation order
        # (of the all-reduce, and of the shards' own weight-gradient kernels) decides its sign and the update differs
        # by a full step (lr 2e-3 ... 5e-3) in that element, every step it stays near zero -- a handful of elements at
        # most, by a few steps at most
        off = np.abs(b - a) > 3e-3 + 5e-3 * np.abs(a)
        assert off.sum() <= max(1, 0.01 * off.size) and np.abs(b - a).max() < 0.03, \
            (k, int(off.sum()), off.size, float(np.abs(b - a).max()))
        assert np.sum(np.abs(b - a) > 5e-4 + 5e-3 * np.abs(a)) <= max(1, 0.02 * off.size), k


def test_infer_inv_gamma_pretraining(tmp_path, monkeypatch):
    """A pre-training run with the learned inverse-gamma hyper-prior (diagonal family): the loss falls, the four
    hyper-parameters move away from their initial (20, 2.5, 20, 2.5) and travel with the checkpoint."""
    from qbold_vi_amd import training
    monkeypatch.chdir(ROOT)
    cfg = small_config(tmp_path, no_pt_epochs=25, no_ft_epochs=0, use_mvg=False, infer_inv_gamma=True)
    params = training.get_params()
    model, trainer, _ = training.create_and_train_on_synthetic_data(cfg, params, log=training.MetricsLog(echo=False),
                                                                    sample_size=200)
    hp = model.hyper_params()
    assert np.isfinite(hp).all() and np.abs(np.log(hp) - np.log([20.0, 2.5, 20.0, 2.5])).max() > 1e-3
    w = model.get_weights()
    assert "hyper_prior" in w and w["hyper_prior"].shape == (4,)
    model2, _, _ = training.create_encoder_model(cfg, params)
    model2.set_weights(w)
    np.testing.assert_allclose(model2.hyper_params(), hp, rtol=1e-6)
    out1 = model.predict(torch.rand(16, 1, 1, 1, 11, device="cuda") + 0.3, want=("out1",))[0]
    assert out1.shape[-1] == 8
