#!/bin/bash
# tools/gpu_ab_celt.sh <variant>...: the new CELT tests on the product build, then the dense CELT batch per library variant
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
( timeout 900 python -m pytest tests/test_celt_walk_gpu.py -x -q 2>&1 | tail -5 )
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so python tools/bench_codecs.py --codec celt --steps 5 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)['celt']; print('$v', round(j['avg_kernel_ms'],3), round(j['frac'],4), j['rms_vs_oracle'], j['int16_flip_rate'])
"; done
