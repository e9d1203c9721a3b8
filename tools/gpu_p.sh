#!/bin/bash
# Vorbis floor on the device: kernel + outer-surface tests, then the end-to-end A/B (host floor vs device floor) and FLAC e2e
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02p; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_vorbis_floor_gpu.py tests/test_vorbis_gpu.py tests/test_stream_gpu.py tests/test_multidevice_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r02p/pytest.log
cat gpurun_out/r02p/pytest.log
for v in 1 ""; do
  echo "AFG_VORBIS_HOST_FLOOR=$v"
  AFG_VORBIS_HOST_FLOOR=$v AFG_TRACE=1 timeout 600 python tools/bench_codecs.py --codec vorbis_e2e --steps 5 --warmup 2 2> gpurun_out/r02p/ve2e_$v.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d=d.get('vorbis_e2e', d)
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))})"
  grep -E "ogg|vorbis" gpurun_out/r02p/ve2e_$v.err | tail -8
done
timeout 600 python tools/bench_codecs.py --codec flac_e2e --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d=d.get('flac_e2e', d)
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))})"
