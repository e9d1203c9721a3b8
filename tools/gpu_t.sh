#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for abl in 0 2; do
AFG_DE_ABL=$abl timeout 900 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --only celt 2> /tmp/c5.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c5 celt only duo default abl=$abl', round(d['ms_per_step'],2), d['parity'])" || tail -5 /tmp/c5.err
done
