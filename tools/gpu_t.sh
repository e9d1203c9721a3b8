#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for r in 0 1; do
AFG_CELT_ROUNDS=$r timeout 600 python tools/bench_codecs.py --codec celt --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['celt']; print('rounds=$r', round(d['avg_kernel_ms'],2), d['frac'], d['bitwise_mismatches'])"
done
AFG_CELT_DE_DUO=0 timeout 600 python tools/bench_codecs.py --codec celt --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['celt']; print('rounds=1 duo=0', round(d['avg_kernel_ms'],2), d['frac'], d['bitwise_mismatches'])"
