#!/bin/bash
# tools/gpu_ab_vorbis3.sh [variant...]: Vorbis tests on the product build, then C3 (bench.py --config c3) for the product build and variants
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
( timeout 900 python -m pytest tests/test_vorbis_gpu.py tests/test_vorbis_floor_gpu.py -m gpu -x -q 2>&1 | tail -3 )
run() { python bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels'][0]; print('$1', round(k['avg_kernel_ms'],3), round(k['frac'],4), d['parity']['vorbis']['mismatches'])
"; }
run product
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so run $v; done
run product
