#!/bin/bash
# tools/gpu_ab_flac.sh [variant...]: C4 (bench.py --config c4) for the product build and each named library variant
# (audio-formats_amd/lib/libafg_<variant>.so), int16 rows; "<variant>:32" runs that variant with int32 rows
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
run() { python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels'][0]; print('$1', round(k['avg_kernel_ms'],3), round(k['frac'],4), d['parity']['flac']['mismatches'])
"; }
run product
for v in "$@"; do
  name=${v%%:*}
  if [ "$name" != "$v" ]; then
    if [ "$name" = product ]; then AFG_FLAC_RES32=1 run product:32; else AFG_FLAC_RES32=1 AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$name.so run $v; fi
  else AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$name.so run $v; fi
done
