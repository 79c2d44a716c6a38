#!/bin/bash
# Ablation builds of the interleaved split forward kernel: tools/scratch/libsynthsr_abl_<mask>.so (csrc/conv_split.hip, SYN_ABL).
#   bash tools/split_ablate.sh 1 7 8 16 32 63 64      then on the GPU box: python tools/split_ablate_run.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/scratch
cd $R/synthsr_amd/csrc
for m in "$@"; do
  ( hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSYN_ABL=$m -c conv_split.hip -o /tmp/cs_abl_$m.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/scratch/libsynthsr_abl_$m.so generator.o unet_pointwise.o ssim.o critic.o \
          conv_bf16.o /tmp/cs_abl_$m.o conv3d.o ) &
done
wait
ls -la $R/tools/scratch/
