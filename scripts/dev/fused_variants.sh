#!/bin/bash
# Timing experiments on the one-launch wide encoder (run on the GPU box): rebuilds wide_fused_kernels.hip with
# each flag set, relinks, runs the config-3 bench.  Variants with QB_FUSED_X_* compute garbage -- timing only.
cd $GRAFT_REPO_ROOT
C=qbold_vi_amd/csrc; O=qbold_vi_amd/_obj
FL="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-function -DQB_FUSED_DEV"
cp qbold_vi_amd/libqbold_hip.so /tmp/lib_orig.so
for v in "$@"; do
  hipcc $FL $v -c $C/wide_fused_kernels.hip -o /tmp/wf_var.o || exit 1
  objs=$(ls $O/*.o | grep -v wide_fused_kernels)
  hipcc --offload-arch=gfx950 -shared -fPIC -o qbold_vi_amd/libqbold_hip.so $objs /tmp/wf_var.o || exit 1
  echo "variant [$v]: $(python ${QB_VARIANT_SCRIPT:-scripts/dev/time_fused.py} 2>&1 | grep -v amdgpu.ids | tail -${QB_VARIANT_LINES:-1} | tr "\n" " ")"
done
cp /tmp/lib_orig.so qbold_vi_amd/libqbold_hip.so
