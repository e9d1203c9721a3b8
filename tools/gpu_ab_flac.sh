#!/bin/bash
# tools/gpu_ab_flac.sh [variant...]: FLAC tests on the product build, then C4 (bench.py --config c4) per library variant
# and for both lane mappings of the product build
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
( timeout 900 python -m pytest tests/test_flac_gpu.py tests/test_stream_gpu.py -m gpu -x -q 2>&1 | tail -4 )
run() { python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels'][0]; print('$1', round(k['avg_kernel_ms'],3), round(k['frac'],4), d['parity']['flac']['mismatches'])
"; }
run lane=subframe
AFG_FLAC_RES32=1 run lane=subframe,int32rows
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so run $v; done
