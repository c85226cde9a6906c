#!/bin/bash
# dev: a second library with ONE source recompiled under extra flags, for in-process-free A/B runs on one box
# usage: scripts/build_variant.sh NAME SOURCE.hip -DFLAG=... ; then MDF_HIP_LIB=$PWD/mdf-net_amd/csrc/build/libmdfnet_hip_NAME.so python ...
# (the variant library lives under csrc/build/ -- git-ignored, travels to the GPU box -- never in the package directory)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
C=$R/mdf-net_amd/csrc
(cd $R/mdf-net_amd && python -m mdfnet_hip.build >/dev/null)
obj=/tmp/variant_${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -I $R/include -I $C -Wno-unused-function "$@" -x hip -c $C/$src -o $obj
objs=$(ls $C/build/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $C/build/libmdfnet_hip_${name}.so $objs $obj
echo built $C/build/libmdfnet_hip_${name}.so
