#!/bin/bash
# debug library: conv_big_split.hip with in-kernel phase stamps (PGV_BIGQ_STAMPS), every other object from the regular build
set -e
cd "$(dirname "$0")/.."
P=preset-gen-vae_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DPGV_BIGQ_STAMPS "$@" -c $P/csrc/conv_big_split.hip -o scratch/conv_big_split_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DPGV_BIGQ_STAMPS "$@" -c $P/csrc/conv_wgrad_split.hip -o scratch/conv_wgrad_split_stamps.o
objs=$(ls $P/build/*.hip.o | grep -v "conv_big_split\|conv_wgrad_split")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libpgv_hip_stamps.so $objs scratch/conv_big_split_stamps.o scratch/conv_wgrad_split_stamps.o
echo built scratch/libpgv_hip_stamps.so
