#!/bin/bash
# Timing experiments (run on the GPU box): rebuild ONE source file of the library with extra flags, relink, run
# a command, restore the library.  scripts/dev/variant.sh <file.hip> "<flags>" "<command>"
cd $GRAFT_REPO_ROOT
C=qbold_vi_amd/csrc; O=qbold_vi_amd/_obj
SRC=$1; FL=$2; CMD=$3
base=$(basename $SRC .hip)
cp qbold_vi_amd/libqbold_hip.so /tmp/lib_orig.so
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-function $FL -c $C/$SRC -o /tmp/var_$base.o || exit 1
objs=$(ls $O/*.o | grep -v "/$base\.")
hipcc --offload-arch=gfx950 -shared -fPIC -o qbold_vi_amd/libqbold_hip.so $objs /tmp/var_$base.o || exit 1
echo "variant [$SRC $FL]:"; bash -c "$CMD"
cp /tmp/lib_orig.so qbold_vi_amd/libqbold_hip.so
