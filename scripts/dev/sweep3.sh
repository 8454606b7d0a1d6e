#!/bin/bash
# A/B timing of build-time options of the config-3 ELBO kernels on the GPU box (variant.sh on elbo_kernels.hip).
cd $GRAFT_REPO_ROOT
for fl in "$@"; do
  bash scripts/dev/variant.sh elbo_kernels.hip "$fl" "python scripts/dev/time_config3.py 8 2>&1 | grep -v amdgpu | head -2 | tr '\n' ' '; echo"
done
