"""qbold_vi_amd -- MI355X (gfx950) implementation of the qBOLD-VI voxel-wise amortised-VI hot path.

Host side: the reference's Python interface (SignalGenerationLayer, create_synthetic_dataset,
EncoderTrainer, ReparamTrickLayer, LogitMVN, load_arguments) on float32 ROCm tensors.
Device side: libqbold_hip.so, hand-written HIP kernels behind the C ABI of include/qbold_hip.h.
Importing this package does not load the library; the first Context does, and fails loudly if it
has not been built.
"""
__all__ = ["SignalGenerationLayer", "create_synthetic_dataset", "EncoderTrainer",
           "ReparamTrickLayer", "LogitMVN", "load_arguments"]


def __getattr__(name):
    if name in ("SignalGenerationLayer", "create_synthetic_dataset"):
        from . import signals
        return getattr(signals, name)
    if name in ("EncoderTrainer", "ReparamTrickLayer", "EncoderModel", "FineTuner"):
        from . import model
        return getattr(model, name)
    if name == "LogitMVN":
        from .logit_mvn import LogitMVN
        return LogitMVN
    if name == "load_arguments":
        from .utils import load_arguments
        return load_arguments
    raise AttributeError(name)
