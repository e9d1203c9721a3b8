#!/bin/bash
# C5 (bench.py --config c5) under the step orders AFG_C5_ORDER knows: the Opus members beside the others (default) or serial, launch orders
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp
run() { python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1'.ljust(28), round(d['ms_per_step'],2), 'ms', round(d['value']/1e11,3), 'e11', {k['codec']: round(k['avg_kernel_ms'],1) for k in d['roofline']['kernels']})
"; }
run default
AFG_C5_ORDER=serial run serial
AFG_C5_ORDER=flac,vorbis,mp3 run flac,vorbis,mp3
AFG_C5_ORDER=vorbis,flac,mp3 run vorbis,flac,mp3
AFG_C5_ORDER=mp3,flac,vorbis run mp3,flac,vorbis
run default
