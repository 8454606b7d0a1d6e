"""Bivariate logit-Normal distribution -- host mirror of the reference's logit_mvn.py (LogitMVN,
:13-100), whose arithmetic is identical to model.py:288-316, 376-447.  Every method is one
libqbold_hip.so call on float32 ROCm tensors."""
import torch


class LogitMVN:
    def __init__(self, context):
        self._ctx = context
        self._oef_range = 0.8
        self._min_oef = 0.04
        self._dbv_range = 0.2
        self._min_dbv = 0.001

    # logit_mvn.py:46-70 -- NEGATIVE log density of observations [..., 2] under params [..., 5]
    def logit_gaussian_mvg_log_prob(self, observations, predicted_params):
        shape = predicted_params.shape[:-1]
        out = self._ctx.logit_mvn_nlogp(observations.reshape(-1, 2)[:, 0:2],
                                        predicted_params.reshape(-1, 5))
        return out.reshape(shape)

    # logit_mvn.py:20-38: ||L^-1 (obs - mean)||^2 with L = [[e^so, 0], [cov, e^sd]]
    @staticmethod
    def squared_whitened_residual(obs, mean, oef_log_std, dbv_log_std, oef_dbv_cov):
        # Host tensors and tensors inside an autograd graph keep the reference's own expression (device-agnostic and
        # differentiable, like logit_mvn.py:20-38); the kernel serves ROCm tensors outside a graph.
        tensors = (obs, mean, oef_log_std, dbv_log_std, oef_dbv_cov)
        if any((not t.is_cuda) or t.requires_grad for t in tensors):
            out_shape = mean.shape[:-1]
            r = obs.reshape(-1, 2) - mean.reshape(-1, 2)
            so, sd, cov = oef_log_std.reshape(-1), dbv_log_std.reshape(-1), oef_dbv_cov.reshape(-1)
            w0 = r[:, 0] * torch.exp(so * -1.0)
            w1 = r[:, 1] * torch.exp(sd * -1.0) + r[:, 0] * (torch.exp(so * -1.0 + sd * -1.0) * cov * -1.0)
            return (w0 * w0 + w1 * w1).reshape(out_shape)
        from . import _lib
        from .ops import _f32, _ptr, _stream
        out_shape = mean.shape[:-1]
        obs, mean = _f32(obs.reshape(-1, 2), "obs"), _f32(mean.reshape(-1, 2), "mean")
        n = mean.shape[0]
        so, sd, cov = (_f32(t.reshape(-1), name) for t, name in ((oef_log_std, "oef_log_std"),
                                                                  (dbv_log_std, "dbv_log_std"), (oef_dbv_cov, "oef_dbv_cov")))
        if obs.shape[0] != n or so.numel() != n or sd.numel() != n or cov.numel() != n:
            raise ValueError("squared_whitened_residual: operands disagree on the number of rows")
        out = torch.empty(n, dtype=torch.float32, device=mean.device)
        _lib.check(_lib.load().qbold_squared_whitened_residual(_ptr(obs), _ptr(mean), _ptr(so), _ptr(sd), _ptr(cov),
                                                               _ptr(out), n, _stream()), "qbold_squared_whitened_residual")
        return out.reshape(out_shape)

    @staticmethod
    def calculate_log_chol_det(oef_log_std, dbv_log_std):  # logit_mvn.py:40-44
        return 2.0 * (oef_log_std + dbv_log_std)

    def forward_transform(self, logits):  # logit_mvn.py:72-78
        return self._ctx.transform("forward_transform", logits)

    def backwards_transform(self, signal, include_logit):  # logit_mvn.py:80-89
        return self._ctx.transform("backwards_transform_logit" if include_logit
                                   else "backwards_transform", signal)

    def transform_std(self, pred_stds):  # logit_mvn.py:91-93
        return self._ctx.transform("transform_std", pred_stds)

    def transform_offdiag(self, pred_offdiag):  # logit_mvn.py:95-97
        return self._ctx.transform("transform_offdiag", pred_offdiag)

    def inv_transform_std(self, std):  # logit_mvn.py:99-100
        return self._ctx.transform("inv_transform_std", std)
