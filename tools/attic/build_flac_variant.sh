#!/bin/bash
# tools/build_flac_variant.sh <name> <flac_restore source> [extra hipcc flags...]
# A/B library with another FLAC restore kernel and the product's other objects:
#   -> audio-formats_amd/lib/libafg_<name>.so (use with AFG_LIB_PATH; development only).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/audio-formats_amd
( cd "$pkg" && make -s )
mkdir -p "$pkg/build/var_$name"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function \
    -I"$pkg/csrc" -I"$root/include" "$@" -x hip -c "$src" -o "$pkg/build/var_$name/flac_restore.o"
objs=$(ls "$pkg"/build/*.o | grep -v '/flac_restore.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$pkg/lib/libafg_$name.so" $objs "$pkg/build/var_$name/flac_restore.o" -lpthread
echo "built $pkg/lib/libafg_$name.so"
