#!/bin/bash
# C3 with the Vorbis walk item length forced (AFG_VORBIS_SEG_PACKETS): the re-read of the packet in front of every item against the tail
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp
run() { python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-others --no-full-fetch 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernels'][0]; print('$1', round(k['avg_kernel_ms'],3), round(k['frac'],4), d['parity']['vorbis'].get('rms_error'))
"; }
run default
for s in 16 32 64 128 256; do AFG_VORBIS_SEG_PACKETS=$s run seg$s; done
run default
