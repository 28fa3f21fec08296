#!/bin/bash
# variant library for A/B timing: scratch/build_variant.sh NAME "-DFLAG ..." file1.hip [file2.hip ...]
# recompiles the named translation units with the extra flags and links them with the product's other objects
# -> scratch/libpgv_NAME.so (use with scratch/time_launches.py --lib scratch/libpgv_NAME.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P="$ROOT/preset-gen-vae_amd"
NAME=$1; FLAGS=$2; shift 2
mkdir -p "$ROOT/scratch/var_$NAME"
(cd "$P" && python build_ext.py >/dev/null 2>&1)
OBJS=""
for o in "$P"/build/*.hip.o; do
  b=$(basename "$o" .o)
  skip=0
  for f in "$@"; do [ "$b" = "$f" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for f in "$@"; do
  EXTRA=""
  [ "$f" = "stft_mel.hip" ] && EXTRA="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $EXTRA $FLAGS -c "$P/csrc/$f" -o "$ROOT/scratch/var_$NAME/$f.o" &
done
wait
for f in "$@"; do OBJS="$OBJS $ROOT/scratch/var_$NAME/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scratch/libpgv_$NAME.so" $OBJS
echo "built scratch/libpgv_$NAME.so"
