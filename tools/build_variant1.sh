#!/bin/bash
# tools/build_variant1.sh <name> <source.hip> [extra hipcc flags...]: A/B variant of ONE kernel source, linked with the
# product's other objects -> audio-formats_amd/lib/libafg_<name>.so (use with AFG_LIB_PATH; development only).
set -e
name=$1; src=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/audio-formats_amd
make -s -C "$pkg" >/dev/null
bd=$pkg/build/var1_$name
mkdir -p "$bd"
base=$(basename "$src" .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function "$@" -c "$pkg/csrc/$base.hip" -o "$bd/$base.o"
objs=$(ls "$pkg"/build/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$pkg/lib/libafg_$name.so" $objs "$bd/$base.o" -lpthread
echo "built libafg_$name.so"
