#!/bin/bash
# tools/gpu_vorbis_walk.sh: the Vorbis walk's tests, then C3 in both numeric modes (tolerance = vorbis_walk.hip)
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
( timeout 900 python -m pytest tests/test_vorbis_walk_gpu.py tests/test_vorbis_gpu.py -m gpu -x -q 2>&1 | tail -15 )
run() { python tools/bench_codecs.py --codec vorbis --steps 5 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)['vorbis']; print('$1', round(j['avg_kernel_ms'],3), round(j['frac'],4), j['bitwise_mismatches'], j['rms_error'])
"; }
AFG_NUMERIC=exact run exact
run tolerance
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so run $v; done
