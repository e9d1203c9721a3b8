#!/bin/bash
# AddressSanitizer / UBSan mutation fuzzing of the host front-ends on the CPU (no device involved; GPU ASan is not
# available on this pool).  usage: tools/fuzz/run.sh [iterations]
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=${TMPDIR:-/tmp}/afg_fuzz
mkdir -p "$out"
n=${1:-2000}
cd "$root/audio-formats_amd/host"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I. "$root/tools/fuzz/fuzz_mp3.cpp" afg_mp3_front.cpp -o "$out/fuzz_mp3"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I. "$root/tools/fuzz/fuzz_flac.cpp" afg_flac_front.cpp -o "$out/fuzz_flac"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I. "$root/tools/fuzz/fuzz_vorbis.cpp" afg_vorbis_front.cpp -o "$out/fuzz_vorbis"
for seed in 1 2 3; do "$out/fuzz_mp3" "$root/tests/golden/mathjax_invalid_keypress.mp3" $seed "$n"; done
for seed in 1 2 3; do "$out/fuzz_vorbis" "$root/tests/golden/mathjax_invalid_keypress.ogg" $seed "$n"; done
cd "$root/tests" && python3 - "$out" <<'PY'
import sys
sys.path.insert(0, "../audio-formats_amd")
import flac_bitstream as fb
from test_flac_frontend import make_pcm
d, _ = fb.encode_file(make_pcm(4096 * 2 + 500, 2, 16, 3), 16, 1024, orders=(8, 12, 3, 32))
open(sys.argv[1] + "/a.flac", "wb").write(d)
d, _ = fb.encode_file(make_pcm(1152 * 3, 2, 24, 4), 24, 1152)
open(sys.argv[1] + "/b.flac", "wb").write(d)
PY
for f in a b; do "$out/fuzz_flac" "$out/$f.flac" 1 "$n"; done
