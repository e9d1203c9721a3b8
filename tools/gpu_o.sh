#!/bin/bash
# int16 FLAC residual rows: tests, then the C4 A/B and the headline bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02o; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_flac_gpu.py tests/test_stream_gpu.py tests/test_multidevice_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r02o/pytest.log
cat gpurun_out/r02o/pytest.log
for v in 1 ""; do
  AFG_FLAC_RES32=$v python bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('res32=$v', [(k['codec'], round(k['avg_kernel_ms'],2), round(k['frac'],3)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c234', '%.4g' % d['value'], round(d['ms_per_step'],2), [(k['codec'], round(k['avg_kernel_ms'],2), round(k['frac'],3)) for k in d['roofline']['kernels']], {k:v['mismatches'] for k,v in d['parity'].items()})"
