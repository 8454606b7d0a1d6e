"""ctypes binding of libqbold_hip.so (the C ABI declared in include/qbold_hip.h).

The library is the product: if it is missing or fails to load this module raises -- there is no
CPU or PyTorch fallback for any arithmetic on the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QBOLD_LIB selects another build of the SAME ABI for a timing experiment (build.build_lib(extra_flags=...) writes
# such builds to their own object directory and library file, never over this one).
LIB_PATH = os.environ.get("QBOLD_LIB") or os.path.join(_HERE, "libqbold_hip.so")

QBOLD_OK = 0
QBOLD_TISSUE_TABLE = 0
QBOLD_TISSUE_LITERAL = 1


class QboldError(RuntimeError):
    pass


class Consts(C.Structure):
    """qbold_consts -- INI `config` [DEFAULT] as parsed at signals.py:18-53 of the reference."""
    _fields_ = [(n, C.c_double) for n in
                ("gamma", "b0", "dchi", "te", "r2t", "tr", "ti", "t1b", "hct",
                 "tau_start", "tau_end", "tau_step")] + \
               [("full_model", C.c_int32), ("include_blood", C.c_int32)]


class LossCfg(C.Structure):
    _fields_ = [("multi_image_normalisation", C.c_int32), ("predict_log_data", C.c_int32),
                ("use_student_t", C.c_int32), ("student_t_df", C.c_double)]


class EncoderShape(C.Structure):
    _fields_ = [("T", C.c_int32), ("U", C.c_int32), ("L", C.c_int32),
                ("channelwise_gating", C.c_int32), ("gate_offset", C.c_float),
                ("spatial_taps", C.c_int32), ("precision", C.c_int32), ("activation", C.c_int32),
                ("layer_norm", C.c_int32), ("dropout_rate", C.c_float), ("dropout_seed", C.c_uint64)]


class Geometry(C.Structure):
    """qbold_geometry: a [B][X][Y][Z][C] crop batch."""
    _fields_ = [("B", C.c_int32), ("X", C.c_int32), ("Y", C.c_int32), ("Z", C.c_int32)]


_P = C.c_void_p
_I64 = C.c_int64
_U64 = C.c_uint64

# name -> (restype, argtypes); the single source of truth used by tests to check the export list
SIGNATURES = {
    "qbold_abi_version": (C.c_int, []),
    "qbold_last_error": (C.c_char_p, []),
    "qbold_ctx_create": (C.c_int, [C.POINTER(Consts), C.POINTER(LossCfg), C.c_int, C.POINTER(_P)]),
    "qbold_ctx_destroy": (None, [_P]),
    "qbold_ctx_num_taus": (C.c_int, [_P]),
    "qbold_ctx_se_idx": (C.c_int, [_P]),
    "qbold_ctx_taus": (C.c_int, [_P, _P]),
    "qbold_ctx_set_tissue_mode": (C.c_int, [_P, C.c_int]),
    "qbold_ctx_tissue_mode": (C.c_int, [_P]),
    "qbold_ctx_set_grad_node0": (C.c_int, [_P, C.c_int]),
    "qbold_ctx_set_kernel_selection": (C.c_int, [_P, C.c_int]),
    "qbold_squared_whitened_residual": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _P]),
    "qbold_ctx_table_eval": (C.c_int, [_P, _P, _P, _P, _I64]),
    "qbold_signal_fwd": (C.c_int, [_P, _P, _P, _I64, _P]),
    "qbold_signal_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    "qbold_encoder_num_params": (_I64, [C.POINTER(EncoderShape)]),
    "qbold_encoder_packed_floats": (_I64, [C.POINTER(EncoderShape)]),
    "qbold_encoder_pack": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, _P]),
    "qbold_encoder_fwd": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, _P, _P, _P, _I64, _P]),
    "qbold_reparam": (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    "qbold_logit_mvn_nlogp": (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    "qbold_posterior_moments": (C.c_int, [_P, _P, _P, C.c_int, _U64, _I64, _P, _P, _I64, _P]),
    "qbold_normalise": (C.c_int, [_P, _P, _P, _I64, _P]),
    "qbold_transform": (C.c_int, [_P, C.c_int, _P, _P, _I64, _P]),
    "qbold_nll_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, C.c_int, _P]),
    "qbold_kl_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int, _U64, _I64, _P, _I64, _P]),
    "qbold_kl_closed": (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    "qbold_normals": (C.c_int, [_P, _U64, C.c_uint32, _I64, C.c_int, _P, _I64, _P]),
    "qbold_noise_workspace_bytes": (_I64, [_P]),
    "qbold_signal_add_noise": (C.c_int, [_P, _P, _P, C.c_float, C.c_float, _U64, _I64, _P, _I64, _P]),
    "qbold_elbo_workspace_bytes": (_I64, [_P]),
    "qbold_elbo_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _U64, _I64,
                                 _P, _P, _P, _I64, _P]),
    "qbold_elbo_fwd_logsigma": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _U64, _I64, _P, _P, _P, _I64, _P]),
    "qbold_elbo_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _U64, _I64, _P, _P, _P, _P,
                                 _P, _I64, _P]),
    "qbold_train_workspace_floats": (_I64, [C.POINTER(EncoderShape), _I64]),
    "qbold_encoder_train_fwd": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, C.c_int, _P, _P, _P, _I64, _P]),
    "qbold_encoder_train_fwd_fused": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, C.c_int, _P, _P, _P, _I64, _P]),
    "qbold_encoder_train_bwd_recomputes": (C.c_int, [_P, C.POINTER(EncoderShape), _I64]),
    "qbold_encoder_train_bwd": (C.c_int, [_P, C.POINTER(EncoderShape), _P, C.c_int, _P, _P, _P, _P, _P,
                                          _I64, _P]),
    "qbold_encoder_spatial_fwd": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, C.POINTER(Geometry), _P, _P,
                                            _P, _P]),
    "qbold_encoder_spatial_bwd": (C.c_int, [_P, C.POINTER(EncoderShape), _P, C.POINTER(Geometry), _P, _P, _P,
                                            _P, _P, _P]),
    "qbold_encoder_wide_packed_floats": (C.c_int64, [C.POINTER(EncoderShape)]),
    "qbold_encoder_wide_workspace_floats": (C.c_int64, [C.POINTER(EncoderShape), C.c_int64]),
    "qbold_encoder_wide_pack": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, _P]),
    "qbold_encoder_wide_fwd": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, C.c_int, _P, _P, _P, C.c_int64, _P]),
    "qbold_encoder_fused_packed_floats": (C.c_int64, [C.POINTER(EncoderShape)]),
    "qbold_encoder_fused_pack": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, _P]),
    "qbold_encoder_fused_fwd": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, _P, _P, C.c_int64, _P]),
    "qbold_vi_workspace_bytes": (C.c_int64, [_P, C.POINTER(EncoderShape), C.c_int64]),
    "qbold_signal_fwd_ex": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "qbold_kl_mog": (C.c_int, [_P, _P, _P, C.c_int, _P, _U64, _I64, _P, _I64, _P]),
    "qbold_kl_diag": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "qbold_wls_fit": (C.c_int, [_P, _P, C.c_double, _P, C.c_int64, _P]),
    "qbold_smoothness": (C.c_int, [_P, _P, _P, C.POINTER(Geometry), C.c_float, _P, _P, _P]),
    "qbold_synth_loss_bwd": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_float, C.c_double, C.c_double, _I64, _P]),
    "qbold_hyper_prior_bwd": (C.c_int, [_P, _P, _P, C.c_float, _P, _P, _P, _I64, _P]),
    "qbold_r2p_loss_bwd": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _U64, _I64, C.c_float, _P, _P, _I64, _P]),
    "qbold_adamw_step": (C.c_int, [_P, _P, _P, _P, _P, _I64, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_double, _I64, _P]),
    "qbold_vi_fwd": (C.c_int, [_P, C.POINTER(EncoderShape), _P, _P, _P, _P, C.c_int, C.c_int,
                               _U64, _I64, _P, _P, _P, _P, _I64, _P]),
}

_lib = None


def load():
    """Load the shared library once; raise QboldError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QboldError(
            f"{LIB_PATH} not found: build it with `python -m qbold_vi_amd.build` "
            "(or __graft_entry__.build()); there is no fallback path")
    # PyTorch ships its own HIP runtime (SONAME libamdhip64.so.7).  It must be the one already
    # loaded when libqbold_hip.so's NEEDED entry is resolved, so that torch's streams and device
    # pointers belong to the runtime the kernels are launched through.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != QBOLD_OK:
        msg = load().qbold_last_error()
        raise QboldError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")
