#!/bin/bash
# tools/gpu_vorbis_walk.sh [variant...]: the Vorbis walk's tests, then C3 in both numeric modes (tolerance = vorbis_walk.hip)
# and for library variants (tools/build_variant1.sh vw_<name> vorbis_walk.hip -ffp-contract=fast ...)
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
( timeout 900 python -m pytest tests/test_vorbis_walk_gpu.py tests/test_vorbis_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -5 )
run() { tag=$1; shift; python tools/bench_codecs.py --codec vorbis --steps 5 "$@" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)['vorbis']; print('$tag', round(j['avg_kernel_ms'],3), round(j['frac'],4), j['bitwise_mismatches'], j['rms_error'])
"; }
AFG_NUMERIC=exact run exact
run tolerance
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_vw_$v.so run $v; done
run tolerance_again
