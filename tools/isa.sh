#!/bin/bash
# tools/isa.sh <csrc/file.hip> [out.s] [extra flags...]: the gfx950 assembly of one translation unit, with the flags the Makefile
# gives it (the tolerance-mode units with -ffp-contract=fast), for reading and for counting instructions
R=$(cd "$(dirname "$0")/.." && pwd)
src=$1; out=${2:-/tmp/$(basename "$src" .hip).s}; shift; shift
fp="-ffp-contract=off"
case "$(basename "$src")" in celt_walk.hip|mp3_tolerance.hip|vorbis_walk.hip) fp="-ffp-contract=fast";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $fp -fno-fast-math -fno-slp-vectorize -Wno-unused-function \
  --cuda-device-only -S -o "$out" "$@" "$R/audio-formats_amd/$src" && echo "$out"
