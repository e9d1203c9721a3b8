#!/bin/bash
# C5's Opus members alone (bench.py --config c5 --only celt) for a few item sizes of the CELT walk
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
for seg in "$@"; do
AFG_CELT_SEG_RECS=$seg python bench.py --config c5 --only celt --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('seg $seg', '%.4g' % d['value'], round(d['ms_per_step'],2), [(k['codec'],round(k['avg_kernel_ms'],2),round(k['frac'],3)) for k in d['roofline']['kernels']], d['parity']['celt']['rms_error'])
"
done
