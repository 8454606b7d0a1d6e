"""Import shim with the reference's module name (qbold_build_model.py:11-82)."""
from qbold_vi_amd.training import ModelBuilder, WeightStatus  # noqa: F401
