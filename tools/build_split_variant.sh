#!/bin/bash
# bash tools/build_split_variant.sh NAME "-DFLAG=..."  -> tools/scratch/lib_NAME.so: ONLY conv_split.hip recompiled with the flags
# (linked against the shipped objects of the other sources; tools/build_variant.sh also rebuilds conv3d.hip)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
C=$R/synthsr_amd/csrc; S=$R/tools/scratch
mkdir -p $S
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $C/conv_split.hip -o $S/conv_split_$N.o "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $S/lib_$N.so $C/generator.o $C/unet_pointwise.o $C/ssim.o $C/critic.o $C/conv_bf16.o $S/conv_split_$N.o $C/conv3d.o
echo $S/lib_$N.so
