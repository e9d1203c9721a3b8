#!/bin/bash
# tools/gpu_e2e_threads.sh: sustained end-to-end rates against the host thread count (the boxes: 256 logical CPUs, 16-CPU quota)
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
for codec in mp3_e2e vorbis_e2e flac_e2e; do
  for t in 0 16 24 32 64; do
    python tools/bench_codecs.py --codec $codec --e2e-threads $t --e2e-distinct 64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        v=list(json.loads(l).values())[0]; print('$codec threads=$t', round(v['samples_per_s_end_to_end']/1e9,2), 'Gs/s', round(v['seconds']*1e3,1), 'ms', 'parity', v['parity']['mismatches'])
"
  done
done
