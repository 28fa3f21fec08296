#!/bin/bash
# debug library: conv_v2.hip with in-kernel phase stamps (PGV_V2_TIMING), every other object from the regular build
# usage: scratch/build_dbg.sh [suffix] [extra -D flags...]   -> scratch/libpgv_hip_dbg<suffix>.so
set -e
cd "$(dirname "$0")/.."
P=preset-gen-vae_amd
sfx=$1; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DPGV_V2_TIMING "$@" -c $P/csrc/conv_v2.hip -o scratch/conv_v2_dbg$sfx.o
objs=$(ls $P/build/*.hip.o | grep -v conv_v2)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libpgv_hip_dbg$sfx.so $objs scratch/conv_v2_dbg$sfx.o
echo built scratch/libpgv_hip_dbg$sfx.so
