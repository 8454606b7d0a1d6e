#!/bin/bash
cd $GRAFT_REPO_ROOT
C=qbold_vi_amd/csrc; O=qbold_vi_amd/_obj
FL="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-function -DQB_FUSED_DEV -DQB_FUSED_STAMP"
cp qbold_vi_amd/libqbold_hip.so /tmp/lib_orig.so
hipcc $FL $@ -c $C/wide_fused_kernels.hip -o /tmp/wf_var.o || exit 1
objs=$(ls $O/*.o | grep -v wide_fused_kernels)
hipcc --offload-arch=gfx950 -shared -fPIC -o qbold_vi_amd/libqbold_hip.so $objs /tmp/wf_var.o || exit 1
timeout -k 10 200 python scripts/dev/stamp_fused.py 2>&1 | tail -16
cp /tmp/lib_orig.so qbold_vi_amd/libqbold_hip.so
