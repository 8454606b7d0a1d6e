"""Encoder, reparameterised sampling and the variational objective -- host mirror of the
reference's model.py (ReparamTrickLayer :15-50, EncoderTrainer :53-770).

Class and method names, argument order and defaults follow the reference; tensors are float32
ROCm tensors, channel-last, with the mask carried as the LAST channel of the "true" tensors
exactly as Keras hands them to the reference's loss callables.  Voxel batches may have any leading
shape; the reference's 5-D (B, X, Y, Z, C) layout is accepted as is.  Arithmetic is done by
libqbold_hip.so; torch provides memory, reshapes and trivial reductions of kernel outputs.

Scope: the branches optimal.yaml disables are built where the reference itself can run them (population
prior / MoG / inverse-gamma with the diagonal family, dropout and GroupNorm on the layer-wise kernels); the pairs the
reference cannot evaluate raise NotImplementedError instead of silently computing something else.  Image crops [B, X, Y, Z, C]
 take the layer-wise spatial kernels; voxel batches the fused ones.
"""
import math

import numpy as np
import torch

from .init import init_encoder_weights
from .ops import Context, EncoderWeights


def _pad5(t):
    """Distribution parameters as the kernels take them: 5 per voxel.  The diagonal family
    (use_mvg=False, model.py:33-37) is the 5-parameter one with a zero Cholesky term."""
    if t.shape[-1] == 5:
        return t
    if t.shape[-1] == 8:   # infer_inv_gamma: [4 parameters | 4 hyper-parameters] (model.py:205): the parameters
        t = t[..., :4]
    if t.shape[-1] != 4:
        raise ValueError("distribution parameters must have 4 (use_mvg=False) or 5 channels")
    return torch.cat([t, torch.zeros_like(t[..., :1])], -1)


def _flat(t, c):
    return t.reshape(-1, c)


R2P_LOSS_SAMPLES = 10   # n_samples of the R2' term, model.py:479

class EncoderModel:
    """Stands in for the Keras `outer_model` of create_encoder (model.py:222): model(x) ->
    [out1 (stream 1, [...,5]), out2 (stream 2, [...,5]), sigma ([...,T])]."""

    def __init__(self, trainer, weights):
        self._trainer = trainer
        self.weights = weights
        # infer_inv_gamma (model.py:201-205): a tfp VariableLayer of four exp-activated scalars -- the inverse-gamma
        # (alpha, beta) of the OEF and DBV variances -- initialised to log([20, 2.5, 20, 2.5]) and appended to the
        # first output as constant channels.  Host state (four numbers): hyper_raw holds the logs.
        self.hyper_raw = np.log(np.array([20.0, 2.5, 20.0, 2.5])) if trainer._infer_inv_gamma else None

    def hyper_params(self):
        return np.exp(self.hyper_raw)

    def __call__(self, x):
        return self.predict(x)

    def _with_hyper(self, o1):
        if o1 is None or self.hyper_raw is None:
            return o1
        h = torch.as_tensor(self.hyper_params(), dtype=o1.dtype, device=o1.device)
        return torch.cat([o1, h.expand(o1.shape[:-1] + (4,))], -1)   # model.py:205

    def predict(self, x, want=("out1", "out2", "sigma")):
        """Voxel batches go through the fused kernels; image crops [B, X, Y, Z, T] (X or Y > 1) get
        stream 2 with its 3x3x1 'same' convolutions (stream 1 is voxel-wise by construction)."""
        ctx = self._trainer._ctx
        nq = self._trainer._nq
        cut = (lambda t: t if t is None or nq == 5 else t[..., :nq].contiguous())
        if not self._trainer._is_spatial(x):
            o1, o2, sg = ctx.encoder_fwd(self.weights, x, want=want)
            return [self._with_hyper(cut(o1)), cut(o2), sg]
        o1 = self._with_hyper(cut(ctx.encoder_fwd(self.weights, x, want=("out1",))[0])) if "out1" in want else None
        o2 = sg = None
        if "out2" in want or "sigma" in want:
            st = self._trainer._spatial_state(self.weights)
            q, ls = st.forward_spatial(x)
            o2 = cut(q.reshape(x.shape[:-1] + (5,))) if "out2" in want else None
            sg = ctx.transform("exp", ls).reshape(x.shape) if "sigma" in want else None
        return [o1, o2, sg]

    # Weights travel as .npz with the canonical tensor names (Keras HDF5 is SURVEY N3).
    def get_weights(self):
        w = self.weights.to_arrays()
        if self._trainer._nq == 4:  # the reference's final layer has 4 outputs (model.py:191-196)
            w["Wf"], w["bf"] = w["Wf"][:, :4].copy(), w["bf"][:4].copy()
        if self.hyper_raw is not None:
            w["hyper_prior"] = np.asarray(self.hyper_raw, np.float32)
        return w

    def set_weights(self, arrays):
        arrays = dict(arrays)
        if "hyper_prior" in arrays:
            hp = np.asarray(arrays.pop("hyper_prior"), np.float64)
            if self.hyper_raw is not None:
                self.hyper_raw = hp
        if np.asarray(arrays["Wf"]).shape[-1] == 4:
            Wf, bf = np.asarray(arrays["Wf"], np.float32), np.asarray(arrays["bf"], np.float32)
            arrays["Wf"] = np.concatenate([Wf, np.zeros_like(Wf[:, :1])], 1)
            arrays["bf"] = np.concatenate([bf, np.zeros(1, np.float32)])
        self.weights.set_from_arrays(arrays)

    def save_weights(self, path):
        if str(path).endswith((".h5", ".hdf5")):   # Keras layout; needs h5py (keras_h5.py)
            from .keras_h5 import save_keras_h5
            return save_keras_h5(path, self.get_weights())
        np.savez(path, **self.get_weights())

    def load_weights(self, path):
        if str(path).endswith((".h5", ".hdf5")):
            from .keras_h5 import load_keras_h5
            return self.set_weights(load_keras_h5(path))
        with np.load(path) as f:
            self.set_weights({k: f[k] for k in f.files})


class _InnerModel:
    """create_encoder also returns the `inner_model` that starts after the first 1x1x1 layer
    (model.py:217); nothing on the hot path calls it."""

    def __call__(self, *a, **k):
        raise NotImplementedError("inner_model (post-first-layer features in) is not part of the "
                                  "accelerated voxel path")


class ReparamTrickLayer:
    """Draws (OEF, DBV) samples from the predicted logit-Normal (model.py:15-50).  The normals
    come from the library's Philox stream (qbold_normals) unless `z` is given."""

    def __init__(self, encoder_trainer):
        self._encoder_trainer = encoder_trainer
        self._draws = 0

    def __call__(self, inputs, z=None):
        return self.call(inputs, z=z)

    def call(self, inputs, z=None, *args, **kwargs):
        input, mask = inputs
        tr = self._encoder_trainer
        # use_mvg=False (model.py:33-37): independent draws = the Cholesky form with a zero off-diagonal
        q = _pad5(_flat(input, input.shape[-1])[:, :tr._nq]).contiguous()
        if z is None:
            z = tr._ctx.normals(q.shape[0], 1, stream_id=0, seed=tr._seed + 104729 * self._draws)
            self._draws += 1
        return tr._ctx.reparam(q, z.reshape(-1, 2)).reshape(input.shape[:-1] + (2,))


class FineTuner:
    """The `full_model` of build_fine_tuner (model.py:239-286): full_model([data, mask]) ->
    {'predictions': q [...,5], 'predicted_images': concat[signal, sigma] [S*...,2T]}."""

    def __init__(self, trainer, encoder_model, signal_generation_layer):
        self._trainer = trainer
        self.encoder_model = encoder_model
        self.signal_generation_layer = signal_generation_layer
        self._rpl = ReparamTrickLayer(trainer)
        # heteroscedastic_noise=False (model.py:277-281): ONE exp-activated scalar, a variable of the fine-tuner
        # (not of the encoder: the weight files do not carry it), initial value log(initial_im_sigma)
        self.log_sigma = None if trainer._heteroscedastic_noise else math.log(float(trainer._initial_im_sigma))
        # use_population_prior with the diagonal family (model.py:262-271): a 4-vector variable of the fine-tuner,
        # [mu_oef, raw_s_oef, mu_dbv, raw_s_dbv], appended to 'predictions' as constant channels
        # (mog_components > 1: 4 M values, random-normal initialised, model.py:262-266)
        M = int(trainer._mog_components)
        self.pop_prior = None
        if trainer._use_population_prior:
            self.pop_prior = (np.array([-0.97, 0.4, -1.14, 0.6], np.float32) if M <= 1 else
                              np.random.default_rng(trainer._seed).standard_normal(4 * M).astype(np.float32))

    def __call__(self, inputs):
        return self.predict(inputs)

    def predict(self, inputs):
        data, mask = inputs
        tr = self._trainer
        _, q, sigma = self.encoder_model.predict(data, want=("out2", "sigma"))
        S = tr._no_samples
        qs = torch.cat([q] * S, 0)          # model.py:245
        sig = torch.cat([sigma] * S, 0)     # model.py:246
        sampled = self._rpl((qs, mask))     # model.py:248
        if self.pop_prior is not None:      # model.py:268-270
            pp = torch.as_tensor(self.pop_prior, device=qs.device)
            qs = torch.cat([qs, pp.expand(qs.shape[:-1] + (pp.numel(),))], -1)
        output = self.signal_generation_layer(sampled)  # model.py:273
        if self.log_sigma is not None:      # model.py:277-281: one channel holding the scalar sigma
            sig = torch.full(output.shape[:-1] + (1,), math.exp(self.log_sigma), dtype=output.dtype,
                             device=output.device)
        # 'predictions' is the S-fold tiled distribution, as in the reference (model.py:245,285)
        return {'predictions': qs, 'predicted_images': torch.cat([output, sig], -1)}

    def elbo(self, data, mask, prior, no_samples=None, kl_samples=70, seed=1, voxel0=0, kl_tiled=None):
        """The fused path: one launch for encoder + S draws + forward model + NLL + K-draw KL
        (image crops: spatial encoder, then the ELBO kernel).
        Returns dict(nll, kl, elbo (= nll + kl, train.py:351), sums, q, nll_kl).

        kl_tiled (default: the trainer's setting, True = the reference's behaviour): with no_samples = S > 1 the
        reference tiles the batch S-fold (model.py:245-246, 656) and kl_loss draws `kl_samples` KL samples per TILED
        row, i.e. S * kl_samples draws per voxel, averaged (model.py:592-610, 661-663).  The kernels never tile; the
        same estimator is S * kl_samples in-kernel draws per voxel.  kl_tiled=False draws kl_samples per voxel
        whatever S (same expectation, S times fewer draws: round 1's behaviour)."""
        tr = self._trainer
        S = tr._no_samples if no_samples is None else no_samples
        kl_samples = tr.kl_draws(kl_samples, S, kl_tiled)
        x = _flat(data, data.shape[-1])
        m = None if mask is None else mask.reshape(-1)
        p5 = _pad5(_flat(prior, prior.shape[-1])).contiguous()
        K = kl_samples if tr._use_mvg else 0   # diagonal family: closed-form KL below (model.py:686-716)
        mog = self.pop_prior is not None and self.pop_prior.size > 4
        if self.pop_prior is not None and not mog:   # the per-voxel prior is ignored (model.py:687-690): one population prior
            p5 = _pad5(torch.as_tensor(self.pop_prior, device=x.device).expand(x.shape[0], 4)).contiguous()
        if tr._is_spatial(data) or self.log_sigma is not None:
            _, q5, sg5 = self.encoder_model.predict(data, want=("out2", "sigma"))
            q = _pad5(_flat(q5, q5.shape[-1])).contiguous()
            sg = _flat(sg5, x.shape[-1])
            if self.log_sigma is not None:   # the encoder's sigma head is not part of this model (model.py:277-281)
                sg = torch.full_like(sg, math.exp(self.log_sigma))
            sums, nll_kl = tr._ctx.elbo_fwd(x, m, q, p5, sg, S, K, seed=seed, voxel0=voxel0)
        else:
            # range_check: activations beyond the f16 operand split's 65504 show as non-finite sums and are
            # recomputed on the exact-float32 layer-wise path (ops.Context.vi_fwd)
            sums, q, nll_kl = tr._ctx.vi_fwd(self.encoder_model.weights, x, m, p5, S, K, seed=seed, voxel0=voxel0,
                                             range_check=True)
        if not tr._use_mvg and mog:   # model.py:666-685: one draw per dimension against the mixture (no prior cost)
            # The reference tiles the batch S-fold before this loss (model.py:245-246, 656), so a voxel is scored at S
            # independent draws and their mean enters the masked sum; copy s of voxel i is row s * N + i of the tiled
            # batch, which is the Philox key kl_loss() gives it.  kl_tiled=False: one draw per voxel.
            comps = torch.as_tensor(self.pop_prior, device=x.device).reshape(-1, 4)
            tiled = tr._kl_tiled if kl_tiled is None else kl_tiled
            copies = max(int(S), 1) if tiled else 1
            kl_v = tr._ctx.kl_mog(q, comps, seed=seed, voxel0=voxel0)
            for s in range(1, copies):
                kl_v = kl_v + tr._ctx.kl_mog(q, comps, seed=seed, voxel0=voxel0 + s * x.shape[0])
            kl_v = kl_v / copies
            live = kl_v if m is None else torch.where(m > 0, kl_v, torch.zeros_like(kl_v))
            sums[1] = live.sum(dtype=torch.float64)
            nll_kl[:, 1] = kl_v
            q = q[:, :4].contiguous()
        elif not tr._use_mvg:
            ksums, kl_v = tr._ctx.kl_diag(q, p5, m)
            sums[1] = ksums[1]
            nll_kl[:, 1] = kl_v
            q = q[:, :4].contiguous()
            if self.pop_prior is not None:   # model.py:710-716: the hyper-prior's cost joins the KL numerator
                sums[1] = sums[1] + tr.population_prior_cost(self.pop_prior, data.shape[0])
        return dict(sums=sums, q=q, nll_kl=nll_kl, nll=sums[0] / sums[2], kl=sums[1] / sums[2],
                    elbo=(sums[0] + sums[1]) / sums[2])


class EncoderTrainer:
    def __init__(self,
                 system_params,
                 no_intermediate_layers=1,
                 no_units=10,
                 use_layer_norm=False,
                 dropout_rate=0.0,
                 activation_type='gelu',
                 student_t_df=None,
                 initial_im_sigma=0.08,
                 multi_image_normalisation=True,
                 channelwise_gating=False,
                 infer_inv_gamma=False,
                 use_mvg=True,
                 use_population_prior=True,
                 mog_components=1,
                 no_samples=1,
                 heteroscedastic_noise=True,
                 predict_log_data=True,
                 full_model=True,
                 use_blood=True,
                 device=None,
                 seed=1):
        self._no_intermediate_layers = no_intermediate_layers
        self._no_units = no_units
        self._use_layer_norm = use_layer_norm
        self._dropout_rate = dropout_rate
        self._activation_type = activation_type
        self._student_t_df = student_t_df
        self._initial_im_sigma = initial_im_sigma
        self._multi_image_normalisation = multi_image_normalisation
        self._system_params = system_params
        self._channelwise_gating = channelwise_gating
        self._infer_inv_gamma = infer_inv_gamma
        self._use_mvg = use_mvg
        self._nq = 5 if use_mvg else 4   # parameters of the predicted distribution (model.py:191-193)
        self._use_population_prior = use_population_prior
        self._mog_components = mog_components
        self._no_samples = no_samples
        self._kl_tiled = True   # KL draws per voxel = no_samples x kl_samples, as the reference's tiled batch gives
        self._oef_range = 0.8
        self._min_oef = 0.04
        self._dbv_range = 0.2
        self._min_dbv = 0.001
        self._heteroscedastic_noise = heteroscedastic_noise
        self._predict_log_data = predict_log_data
        # Store the spin-echo index (model.py:95)
        self._se_idx = int(abs(float(system_params['tau_start']) / float(system_params['tau_step'])))
        self._seed = int(seed)
        unsupported = []
        if activation_type not in ('relu', 'gelu'):
            unsupported.append(f"activation_type={activation_type!r} (kernels implement 'relu' and 'gelu')")
        if infer_inv_gamma and use_mvg:
            # the reference itself cannot run this pair: synthetic_data_loss splits the 9-channel first output
            # (5 + 4) in two (tf.split(y_pred_orig, 2, axis=-1), model.py:455)
            unsupported.append("infer_inv_gamma with use_mvg=True (a shape error in the reference, model.py:455; "
                               "the learned hyper-prior runs with the diagonal family, use_mvg=False)")
        if use_population_prior and use_mvg:
            # the reference cannot run this pair either: kl_loss hands the 10-channel 'predictions' to
            # logit_gaussian_mvg_log_prob, which reshapes them to (-1, 5) and so doubles the rows (model.py:596, 378)
            unsupported.append("use_population_prior with use_mvg=True (a shape error in the reference, model.py:596; "
                               "the population prior runs with the diagonal family, use_mvg=False)")
        if use_population_prior and not use_mvg and mog_components > 16:
            unsupported.append("mog_components > 16")
        if unsupported:
            raise NotImplementedError("configuration outside the accelerated path (disabled by "
                                      "configurations/optimal.yaml): " + "; ".join(unsupported))
        self._ctx = Context(system_params, full_model, use_blood,
                            multi_image_normalisation=multi_image_normalisation,
                            predict_log_data=predict_log_data, student_t_df=student_t_df,
                            device=device)
        assert self._ctx.se_idx == self._se_idx

    # ------------------------------------------------------------------------------------
    @property
    def context(self):
        return self._ctx

    @staticmethod
    def _is_spatial(x):
        return x.dim() == 5 and (x.shape[1] > 1 or x.shape[2] > 1)

    def _spatial_state(self, weights):
        from .ops import TrainState
        st = getattr(weights, "_spatial_state", None)
        if st is None:
            if weights.shape.spatial_taps != 9:
                raise ValueError("image crops need an encoder created with 3x3x1 kernels (spatial_taps=9)")
            st = weights._spatial_state = TrainState(self._ctx, weights, optimiser_state=False)
        return st

    def normalise_data(self, _data):  # model.py:97-113
        return self._ctx.normalise(_data)

    def create_encoder(self, system_constants=None, gate_offset=0.0, resid_init_std=1e-1,
                       no_ip_images=11):
        """model.py:122-223.  Returns (outer_model, inner_model)."""
        if no_ip_images != self._ctx.T:
            raise ValueError("no_ip_images must equal the number of taus of the system parameters")
        # the residual stream owns full 3x3x1 kernels as in the reference (146,176 parameters at
        # optimal.yaml); voxel batches act through their centre tap
        w = init_encoder_weights(T=no_ip_images, U=self._no_units, L=self._no_intermediate_layers,
                                 channelwise_gating=self._channelwise_gating,
                                 resid_init_std=resid_init_std, im_loss_sigma=self._initial_im_sigma,
                                 seed=self._seed, spatial_taps=9, layer_norm=self._use_layer_norm)
        if not self._use_mvg:  # 4 outputs: the Cholesky column of the 5-wide head stays exactly zero
            w["Wf"][:, 4] = 0.0
            w["bf"][4] = 0.0
        ew = EncoderWeights(self._ctx, no_ip_images, self._no_units, self._no_intermediate_layers,
                            self._channelwise_gating, gate_offset, spatial_taps=9,
                            activation=self._activation_type, layer_norm=self._use_layer_norm,
                            dropout_rate=self._dropout_rate).set_from_arrays(w)
        return EncoderModel(self, ew), _InnerModel()

    def build_fine_tuner(self, encoder_model, signal_generation_layer, input_im=None, input_mask=None):
        return FineTuner(self, encoder_model, signal_generation_layer)  # model.py:239-286

    # -- transforms (model.py:288-316) -----------------------------------------------------
    def transform_std(self, pred_stds):
        return self._ctx.transform("transform_std", pred_stds)

    def transform_offdiag(self, pred_offdiag):
        return self._ctx.transform("transform_offdiag", pred_offdiag)

    def inv_transform_std(self, std):
        return self._ctx.transform("inv_transform_std", std)

    def forward_transform(self, logits):
        return self._ctx.transform("forward_transform", logits)

    def backwards_transform(self, signal, include_logit):
        return self._ctx.transform("backwards_transform_logit" if include_logit
                                   else "backwards_transform", signal)

    # -- sampling / moments (model.py:318-374) -----------------------------------------------
    def kl_draws(self, kl_samples, no_samples=None, kl_tiled=None):
        """KL draws per voxel: kl_samples per copy of the S-fold tiled batch of the reference (model.py:245-246, 656),
        i.e. S * kl_samples, unless kl_tiled is off."""
        S = self._no_samples if no_samples is None else no_samples
        tiled = self._kl_tiled if kl_tiled is None else kl_tiled
        return int(kl_samples) * (int(S) if tiled else 1)

    def create_samples(self, predicted_params, mask, no_samples, seed=None):
        q = _pad5(_flat(predicted_params, predicted_params.shape[-1])[:, :self._nq]).contiguous()
        z = self._ctx.normals(q.shape[0], no_samples, stream_id=2,
                              seed=self._seed if seed is None else seed)
        qs = q[:, None, :].expand(-1, no_samples, -1).reshape(-1, 5)
        s = self._ctx.reparam(qs, z.reshape(-1, 2)).reshape(q.shape[0], no_samples, 2)
        return s.permute(0, 2, 1).reshape(predicted_params.shape[:-1] + (2, no_samples))

    def calculate_means(self, predicted_params, mask, include_r2p=False, return_stds=False,
                        no_samples=20, seed=None):
        q = _pad5(_flat(predicted_params, predicted_params.shape[-1])[:, :self._nq]).contiguous()
        means, var = self._ctx.posterior_moments(q, no_samples, seed=self._seed if seed is None else seed,
                                                 want_vars=return_stds)
        c = 3 if include_r2p else 2
        lead = predicted_params.shape[:-1]
        means = means[:, :c].reshape(lead + (c,))
        if return_stds:  # NB: biased variances under the name "stds" (model.py:331, Appendix B9)
            return means, var[:, :c].reshape(lead + (c,))
        return means

    def oef_dbv_metrics(self, y_true, y_pred, oef_dbv_r2p=0):
        means = self.calculate_means(y_pred, None, include_r2p=True)
        residual = (means.reshape(-1, 3) - y_true.reshape(-1, 3))[:, oef_dbv_r2p]
        return torch.mean(residual * residual)

    def oef_metric(self, y_true, y_pred):
        return self.oef_dbv_metrics(y_true, y_pred, 0)

    def dbv_metric(self, y_true, y_pred):
        return self.oef_dbv_metrics(y_true, y_pred, 1)

    def r2p_metric(self, y_true, y_pred):
        return self.oef_dbv_metrics(y_true, y_pred, 2)

    # -- log density (model.py:376-447) ------------------------------------------------------
    def logit_gaussian_mvg_log_prob(self, observations, predicted_params):
        shape = predicted_params.shape[:-1]
        out = self._ctx.logit_mvn_nlogp(observations.reshape(-1, observations.shape[-1])[:, 0:2],
                                        _flat(predicted_params, 5))
        return out.reshape(shape)

    @staticmethod
    def calculate_log_chol_det(oef_log_std, dbv_log_std):
        return 2.0 * (oef_log_std + dbv_log_std)

    @staticmethod
    def squared_whitened_residual(obs, mean, oef_log_std, dbv_log_std, oef_dbv_cov):   # model.py:423-441
        from .logit_mvn import LogitMVN
        return LogitMVN.squared_whitened_residual(obs, mean, oef_log_std, dbv_log_std, oef_dbv_cov)

    def logit_gaussian_log_prob(self, observations, predicted_params):
        """Diagonal family (model.py:406-421): the 5-parameter density with a zero Cholesky term, less
        the log 2 pi that the reference's gaussian_nll (:403-404) does not carry."""
        shape = predicted_params.shape[:-1]
        p = _pad5(_flat(predicted_params, predicted_params.shape[-1])[:, :4]).contiguous()
        out = self._ctx.logit_mvn_nlogp(observations.reshape(-1, observations.shape[-1])[:, 0:2], p)
        return (out - 1.8378770664093453).reshape(shape)

    def synthetic_data_loss(self, y_true_orig, y_pred_orig, use_r2p_loss=False, inv_gamma_alpha=0.0,
                            inv_gamma_beta=0.0):
        """Pre-training loss (model.py:449-514): mean negative log density of the true (OEF, DBV),
        plus the inverse-gamma prior on the marginal variances when alpha * beta > 0 (:492-507)."""
        y = y_true_orig.reshape(-1, 3).contiguous()
        hyper = None
        if self._infer_inv_gamma:   # y_pred = [4 parameters | 4 exp-activated hyper-parameters], model.py:454-455
            flat = _flat(y_pred_orig, y_pred_orig.shape[-1])
            if flat.shape[-1] != 8:
                raise ValueError("infer_inv_gamma: y_pred must carry 4 + 4 channels (model.py:201-205)")
            hyper = [float(v) for v in flat[0, 4:8].tolist()]   # inv_gamma_params[0,0,0,0,:], model.py:494
            y_pred_orig = flat[:, :4]
        q = _pad5(_flat(y_pred_orig, y_pred_orig.shape[-1])[:, :self._nq]).contiguous()
        offset = 0.0 if self._use_mvg else 1.8378770664093453   # logit_gaussian_log_prob, model.py:470
        if hyper is not None:   # model.py:493-498: with infer_inv_gamma the learned prior is the ONLY one
            inv_gamma_alpha = inv_gamma_beta = 0.0
        if inv_gamma_alpha * inv_gamma_beta > 0.0:
            lv = self._ctx.synth_loss(y, q, inv_gamma_alpha, inv_gamma_beta)
        else:
            lv = self._ctx.logit_mvn_nlogp(y[:, :2], q)
        if hyper is not None:   # - log IG(exp(2 s_o); a_o, b_o) - log IG(exp(2 s_d); a_d, b_d), model.py:495-507
            lv = lv.clone()
            self._ctx.hyper_prior_bwd(q, hyper, loss_v=lv, want_stats=False)
        if use_r2p_loss:   # model.py:475-490: ten reparameterised draws, a normal fitted to their R2'
            self._r2p_calls = getattr(self, "_r2p_calls", 0) + 1
            lv = lv.clone()
            self._ctx.r2p_loss_bwd(y, q, R2P_LOSS_SAMPLES, seed=self._r2p_calls, loss_v=lv, want_grad=False)
        return lv.mean() - offset

    def calculate_dw(self, oef):  # model.py:516-522
        from .signals import SignalGenerationLayer
        p = self._system_params
        return SignalGenerationLayer.calculate_dw_static(oef, float(p['hct']), float(p['gamma']),
                                                         float(p['b0']), float(p['dchi']))

    def calculate_r2p(self, oef, dbv):
        return self.calculate_dw(oef) * dbv

    # -- objective terms ---------------------------------------------------------------------
    def fine_tune_loss_fn(self, y_true, y_pred, return_mean=True):
        """model.py:527-568.  y_true [..., T+1] = [data, mask]; y_pred [S*..., 2T] = [signal, sigma], or
        [S*..., T+1] = [signal, one channel of the scalar sigma] with heteroscedastic_noise=False (:535-537)."""
        T = self._ctx.T
        yt = _flat(y_true, T + 1)
        yp = _flat(y_pred, 2 * T if self._heteroscedastic_noise else T + 1)
        S = self._no_samples
        N = yt.shape[0]
        if yp.shape[0] != N * S:
            raise ValueError("y_pred must hold no_samples copies of the batch")
        mask = yt[:, T].contiguous()
        if self._heteroscedastic_noise:
            sig = yp[:, T:]
        else:   # sigma = reduce_mean(y_pred[..., -1:]) (model.py:536): the mean of the constant channel
            sig = yp[:, T:].mean().expand(yp.shape[0], T).contiguous()
        nll = self._ctx.nll_fwd(yt[:, :T], mask, yp[:, :T], sig, S)
        nll = nll * mask.repeat(S)                       # model.py:564
        if return_mean:
            return nll.sum() / (mask.sum() * S)          # model.py:566 (mask is tiled S times)
        return nll.reshape((S * y_true.shape[0],) + tuple(y_true.shape[1:-1]) + (1,))

    def mvg_kl_samples(self, prior, pred, no_samples=50, seed=None):  # model.py:592-610
        pr = _flat(prior, 6)
        kl = self._ctx.kl_fwd(_flat(pred, 5), pr[:, :5], K=no_samples,
                              seed=self._seed if seed is None else seed)
        return kl.reshape(pred.shape[:-1] + (1,))

    def mvg_kl(self, true, predicted):  # model.py:612-652 (closed form; cross-check)
        pr = _flat(true, 6)
        return self._ctx.kl_closed(_flat(predicted, 5), pr[:, :5]).reshape(predicted.shape[:-1] + (1,))

    def kl_loss(self, true, predicted, return_mean=True, no_samples=70, seed=None):  # model.py:654-724
        true = torch.cat([true] * self._no_samples, 0)
        if not self._use_mvg:  # closed form per dimension, model.py:686-721
            pr = _flat(true, 5)   # [p_oef_mean, p_oef_log_std, p_dbv_mean, p_dbv_log_std, mask]
            pred = _flat(predicted, predicted.shape[-1])
            prior_cost = 0.0
            if self._use_population_prior and self._mog_components > 1:   # model.py:666-685
                M = int(self._mog_components)
                if pred.shape[-1] != 4 + 4 * M:
                    raise ValueError("mog_components: predictions must carry 4 + 4 M channels (model.py:262-270)")
                kl = self._ctx.kl_mog(_pad5(pred[:, :4]).contiguous(), pred[0, 4:].reshape(M, 4).contiguous(),
                                      seed=self._seed if seed is None else seed)
                kl_op = kl.reshape(predicted.shape[:-1] + (1,))
                mask = true[..., 4:5]
                kl_op = torch.where(mask > 0, kl_op, torch.zeros_like(kl_op))
                return kl_op.sum() / mask.sum() if return_mean else kl_op
            if self._use_population_prior:   # 'predictions' = [q4 | population prior 4]; `true` gives the mask only
                if pred.shape[-1] != 8:
                    raise ValueError("use_population_prior: predictions must carry 4 + 4 channels (model.py:268-270)")
                prior4 = pred[:, 4:8]
                prior_cost = self.population_prior_cost(prior4[0], predicted.shape[0])
            else:
                prior4 = pr[:, :4]
            _, kl = self._ctx.kl_diag(_pad5(pred[:, :4]).contiguous(), _pad5(prior4).contiguous())
            kl_op = kl.reshape(predicted.shape[:-1] + (1,))
            mask = true[..., 4:5]
            kl_op = torch.where(mask > 0, kl_op, torch.zeros_like(kl_op))
            return (kl_op.sum() + prior_cost) / mask.sum() if return_mean else kl_op
        kl_op = self.mvg_kl_samples(true, predicted, no_samples=no_samples, seed=seed)
        mask = true[..., 5:6]
        kl_op = torch.where(mask > 0, kl_op, torch.zeros_like(kl_op))
        if return_mean:
            return kl_op.sum() / mask.sum()
        return kl_op

    def population_prior_cost(self, prior4, batch):
        """model.py:710-716: -log IG(1, 2) of exp(2 * log-std) for the population prior's DBV and OEF log-stds (after
        transform_std), times the size of the batch axis (tf.shape(predicted)[0]).  Four scalars: host float64."""
        p = [float(v) for v in (prior4.tolist() if hasattr(prior4, "tolist") else prior4)]
        cost = 0.0
        for raw in (p[3], p[1]):
            log_std = 3.0 * math.tanh(raw) - 1.0
            v = math.exp(2.0 * log_std)
            cost -= math.log(2.0) - 2.0 * math.log(v) - 2.0 / v     # log IG(v; 1, 2)
        return cost * float(batch)

    def smoothness_loss(self, true_params, pred_params):
        """Total-variation term (model.py:726-754): sum of |differences| of the range-scaled
        forward-transformed means over x / y neighbours with both masks set, over sum(mask).
        Identically 0 for voxel batches."""
        true_params = torch.cat([true_params] * self._no_samples, 0)
        if not self._is_spatial(pred_params):
            return torch.zeros((), dtype=torch.float32, device=pred_params.device)
        mask = true_params[..., self._nq]   # model.py:729-734
        tv = self._ctx.smoothness(_pad5(pred_params[..., :self._nq]).contiguous(), mask)
        return (tv[0] / mask.sum()).float()

    def estimate_population_param_distribution(self, model, data):
        """model.py:756-770: masked population mean / std of the predicted OEF and DBV logits of the SECOND
        output (`_, predictions, _ = model.predict(...)`, model.py:757: the stream-2 / fine-tuned head, which
        sees the 3x3x1 context on volumes), printed and returned as (mean_oef, log_std_oef, mean_dbv,
        log_std_dbv)."""
        predictions = model.predict(data[..., :-1] * data[..., -1:], want=("out2",))[1]
        mask = data[..., -1:]
        oef, dbv = predictions[..., 0:1] * mask, predictions[..., 2:3] * mask
        mask_pix = mask.sum()
        out = []
        for v in (oef, dbv):
            mean = v.sum() / mask_pix
            std = torch.sqrt((torch.square(v - mean) * mask).sum() / mask_pix)
            out += [mean, self.inv_transform_std(torch.log(std).reshape(1))[0]]
        print('final results for mean_oef, log_std_oef, mean_dbv, log_std_dbv, respectively: ')
        print(*[float(o) for o in out])
        return tuple(out)

    def save_predictions(self, model, data, filename, transform_directory=None, use_first_op=True,
                         fine_tuner_model=None, priors=None):
        """model.py:772-887: write `<filename>_{oef,dbv,r2p,logstds}.nii.gz` (posterior means of
        OEF / DBV / R2' over 200 draws and their variances) and, with a fine tuner,
        `_likelihood` (per-voxel NLL averaged over 100 stochastic passes), `_kl` (100-draw KL to
        `priors`) and `_residual` (mean |normalised data - one sampled prediction|).
        data [subj, X, Y, Z, T+1] with the mask last; each map is stored as [X, Y, Z, subj*C].
        `transform_directory/example.nii.gz`, when present, donates its header (:794-797); the
        FSL `applywarp`/`fslmerge` MNI step (:850-879) is preprocessing outside this package and
        is skipped."""
        import os
        from . import nifti
        data = torch.as_tensor(data, dtype=torch.float32, device=self._ctx.device)
        T = self._ctx.T
        mask = data[..., -1:]
        predictions, predictions2, _ = model.predict(data[..., :-1] * mask)
        if use_first_op is False:
            predictions = predictions2
        means, log_stds = self.calculate_means(predictions, torch.ones_like(predictions[..., :1]),
                                               include_r2p=True, return_stds=True, no_samples=200)
        template = None
        if transform_directory is not None:
            ex = os.path.join(transform_directory, 'example.nii.gz')
            if os.path.isfile(ex):
                template = nifti.load(ex)[1]

        def save_im_data(im_data, _filename):
            im = im_data.detach().cpu().numpy() if torch.is_tensor(im_data) else np.asarray(im_data)
            images = np.concatenate(np.split(im, im.shape[0], axis=0), axis=-1)[0]
            nifti.save(images.astype(np.float32), _filename + '.nii.gz', template)

        if fine_tuner_model:
            if priors is None:
                raise ValueError("save_predictions with a fine tuner needs the prior maps (train.py:272-279)")
            no_passes = 100
            out = fine_tuner_model.elbo(data[..., :-1], mask, torch.as_tensor(priors, device=data.device)[..., :self._nq],
                                        no_samples=no_passes * self._no_samples, kl_samples=100,
                                        seed=self._seed + 17)
            m = mask.reshape(-1)
            lik = (out["nll_kl"][:, 0] * m).reshape(data.shape[:4] + (1,))
            kl = torch.where(m > 0, out["nll_kl"][:, 1], torch.zeros_like(m)).reshape(data.shape[:4] + (1,))
            save_im_data(lik, filename + '_likelihood')
            save_im_data(kl, filename + '_kl')
            y_true = data[..., :-1]
            y_pred = fine_tuner_model.predict([y_true, mask])['predicted_images'][:data.shape[0], ..., :T]
            if self._multi_image_normalisation:
                sl = slice(self._se_idx - 1, self._se_idx + 2)
            else:
                sl = slice(self._se_idx, self._se_idx + 1)
            y_true = y_true / (y_true[..., sl].mean(-1, keepdim=True) + 1e-3)
            y_pred = y_pred / (y_pred[..., sl].mean(-1, keepdim=True) + 1e-3)
            save_im_data((y_true - y_pred).abs().mean(-1, keepdim=True), filename + '_residual')

        save_im_data(means[..., 0:1], filename + '_oef')
        save_im_data(means[..., 1:2], filename + '_dbv')
        save_im_data(means[..., 2:3], filename + '_r2p')
        save_im_data(log_stds, filename + '_logstds')
