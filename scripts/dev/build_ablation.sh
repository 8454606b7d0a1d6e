#!/bin/bash
# Builds an ABLATION copy of the library: -DQBOLD_ABLATION compiles the work-skipping hooks of the timing
# experiments (MEASUREMENTS.md 4.4 / 4.5 / 4.7) in, which then honour QBOLD_DEBUG_SKIP together with
# QBOLD_ALLOW_ABLATION=1.  Never what the tests, the driver or a user load: the default build has no such hook.
# It is written to its own object directory and library file (qbold_vi_amd/libqbold_hip_var_<tag>.so, printed below)
# and is loaded only when QBOLD_LIB names it; qbold_vi_amd/libqbold_hip.so is never touched:
#   L=$(scripts/dev/build_ablation.sh) && QBOLD_LIB=$L QBOLD_ALLOW_ABLATION=1 QBOLD_DEBUG_SKIP=2 python3 bench.py --no_variants
set -e
cd "$(dirname "$0")/../.."
python3 - <<'PY'
from qbold_vi_amd.build import build_lib
print(build_lib(force=True, verbose=False, extra_flags=("-DQBOLD_ABLATION",)))
PY
