#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for v in hip s_max-ilp s_max-memory-clause s_iterative-minreg; do
AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-full-fetch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', '%.4g' % d['value'], round(d['ms_per_step'],2), [(k['codec'], round(k['avg_kernel_ms'],2)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
done
