#!/bin/bash
# Build an A/B variant of the product library: tools/build_variant.sh <name> [extra hipcc flags...]
# -> audio-formats_amd/lib/libafg_<name>.so (use with AFG_LIB_PATH; development only).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/audio-formats_amd
bd=$pkg/build/var_$name
mkdir -p "$bd"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function"
pids=()
for f in "$pkg"/csrc/*.hip; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c "$f" -o "$bd/$(basename "$f" .hip).o" & pids+=($!)
done
for f in "$pkg"/host/*.cpp; do
  /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -c "$f" -o "$bd/host_$(basename "$f" .cpp).o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$pkg/lib/libafg_$name.so" "$bd"/*.o -lpthread
echo "built $pkg/lib/libafg_$name.so"
