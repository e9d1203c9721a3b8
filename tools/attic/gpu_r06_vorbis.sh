#!/bin/bash
# round 6: the channel-packed Vorbis walk -- its tests, then C3 for the product and the named variants, then the other shapes
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_vorbis_walk_gpu.py tests/test_vorbis_gpu.py -m gpu -x -q -n 4 2>&1 | tail -5 )
run() { python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-others --no-full-fetch 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels'][0]; print('$1', round(k['avg_kernel_ms'],3), round(k['frac'],4), d['parity']['vorbis'])
"; }
run product
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so run $v; done
run product
python tools/vorbis_shapes.py --steps 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{') and 'channels' in l and 'vorbis_shapes' not in l:
        d=json.loads(l); print(d['channels'], d['blocksize_0'], d['blocksize_1'], round(d['avg_kernel_ms'],3), round(d['frac'],3), d['rms_error'])
"
