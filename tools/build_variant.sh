#!/bin/bash
# bash tools/build_variant.sh NAME "-DFLAG=..."  -> tools/scratch/lib_NAME.so (conv3d.hip + conv_split.hip recompiled with the flags)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
C=$R/synthsr_amd/csrc; S=$R/tools/scratch
for f in conv3d conv_split; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $C/$f.hip -o $S/${f}_$N.o "$@" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $S/lib_$N.so $C/generator.o $C/unet_pointwise.o $C/ssim.o $C/critic.o $C/conv_bf16.o $S/conv_split_$N.o $S/conv3d_$N.o
echo $S/lib_$N.so
