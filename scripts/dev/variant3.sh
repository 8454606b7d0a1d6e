#!/bin/bash
# Like variant.sh for a flag that several translation units must agree on: rebuild ctx / vi / elbo kernels with
# extra flags, relink, run a command, restore the library.  scripts/dev/variant3.sh "<flags>" "<command>"
cd $GRAFT_REPO_ROOT
C=qbold_vi_amd/csrc; O=qbold_vi_amd/_obj
FL=$1; CMD=$2
cp qbold_vi_amd/libqbold_hip.so /tmp/lib_orig.so
for base in ctx vi_kernels elbo_kernels; do
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-function $FL -c $C/$base.hip -o /tmp/var_$base.o || exit 1 &
done
wait
objs=$(ls $O/*.o | grep -v "/ctx\.\|/vi_kernels\.\|/elbo_kernels\.")
hipcc --offload-arch=gfx950 -shared -fPIC -o qbold_vi_amd/libqbold_hip.so $objs /tmp/var_ctx.o /tmp/var_vi_kernels.o /tmp/var_elbo_kernels.o || exit 1
echo "variant [$FL]:"; bash -c "$CMD"
cp /tmp/lib_orig.so qbold_vi_amd/libqbold_hip.so
