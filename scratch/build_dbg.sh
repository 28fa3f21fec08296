#!/bin/bash
# debug library with in-kernel phase stamps (never shipped): scratch/libpgv_hip_dbg.so
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-unused-value -DPGV_PHASE_TIMING $PGV_DBG_FLAGS \
  preset-gen-vae_amd/csrc/*.hip -o scratch/libpgv_hip_dbg.so
