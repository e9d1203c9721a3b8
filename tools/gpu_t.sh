#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for v in t64; do
AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v c4', [(k['codec'], round(k['avg_kernel_ms'],2), round(k['frac'],3)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
AFG_FLAC_RES32=1 AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v c4 int32 rows', [(k['codec'], round(k['avg_kernel_ms'],2), round(k['frac'],3)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
done
