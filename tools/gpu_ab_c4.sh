#!/bin/bash
# tools/gpu_ab_c4.sh [variant...]: C4 (bench.py --config c4) for the product build and each named library variant
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp
run() { python bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels'][0]; print('$1', round(k['avg_kernel_ms'],3), round(k['frac'],4), d['parity']['flac']['mismatches'])
"; }
run product
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so run $v; done
run product
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so run $v; done
