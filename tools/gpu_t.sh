#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for path in split stream; do
AFG_CELT_PATH=$path timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --c5-wave-files 24576 2> /tmp/c5.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c5 path=$path', '%.4g' % d['value'], round(d['ms_per_step'],2), [(k['codec'], round(k['avg_kernel_ms'],2), round(k['frac'],3)) for k in d['roofline']['kernels']])" || tail -5 /tmp/c5.err
done
