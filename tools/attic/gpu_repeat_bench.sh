#!/bin/bash
# the default bench line several times in a row on one box: run-to-run spread of every kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for i in $(seq 1 ${1:-4}); do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-fetch --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('run $i', '%.4g' % d['value'], round(d['ms_per_step'],2), [(k['codec'], round(k['avg_kernel_ms'],2)) for k in d['roofline']['kernels']])"
done
