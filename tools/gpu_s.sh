#!/bin/bash
# FLAC variant mask + two streams: tests, C4 A/B, headline, e2e
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02s; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_flac_gpu.py tests/test_stream_gpu.py tests/test_multidevice_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r02s/pytest.log
cat gpurun_out/r02s/pytest.log
for v in 1 ""; do
  AFG_FLAC_ALL_VARIANTS=$v python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('all_variants=$v', [(k['codec'], round(k['avg_kernel_ms'],2), round(k['frac'],3)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c234', '%.4g' % d['value'], round(d['ms_per_step'],2), [(k['codec'], round(k['avg_kernel_ms'],2), round(k['frac'],3)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c5', '%.4g' % d['value'], round(d['ms_per_step'],2), [(k['codec'], round(k['avg_kernel_ms'],2), round(k['frac'],3)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
