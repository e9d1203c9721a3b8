#!/bin/bash
# round 6: the multi-channel Vorbis walk (a workgroup per segment, frames put together in LDS): tests, then the >2-channel shapes
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_vorbis_walk_gpu.py tests/test_vorbis_gpu.py -m gpu -x -q -n 4 2>&1 | tail -8 )
timeout 600 python tools/vorbis_shapes.py --steps 5 --only 0,7,8,9 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{') and 'channels' in l and 'vorbis_shapes' not in l:
        d=json.loads(l); print(d['channels'], d['blocksize_0'], d['blocksize_1'], round(d['avg_kernel_ms'],3), round(d['frac'],3), d['rms_error'])
    elif not l.startswith('{'): print(l.rstrip()[:300])
"
