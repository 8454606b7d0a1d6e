#!/bin/bash
# A/B timing of build-time options of the training kernels on the GPU box (variant.sh on train_kernels.hip): each
# argument is one flag set; prints the crop step (and with VOX=1 the voxel step) of scripts/bench_train.py.
cd $GRAFT_REPO_ROOT
ONLY=${ONLY:-crop}
for fl in "$@"; do
  bash scripts/dev/variant.sh train_kernels.hip "$fl" "python scripts/bench_train.py --only $ONLY --steps 30 2>&1 | grep -v amdgpu | tail -1"
done
