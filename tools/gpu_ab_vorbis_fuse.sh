#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vorbis_gpu.py tests/test_golden.py tests/test_stream_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in f12only hip f12only hip; do
AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', [(k['codec'], round(k['avg_kernel_ms'],3), round(k['frac'],3)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
done
