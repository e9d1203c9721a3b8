#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_celt_gpu.py tests/test_multidevice_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r02c/pytest.log
( timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/r02c/c5.err ) > gpurun_out/r02c/c5.json
( AFG_CELT_PATH=stream timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/r02c/c5s.err ) > gpurun_out/r02c/c5_stream.json
( timeout 600 python tools/bench_codecs.py --codec celt --steps 5 --warmup 2 2>&1 | tail -3 ) > gpurun_out/r02c/celt.json
cat gpurun_out/r02c/pytest.log; tail -c 400 gpurun_out/r02c/c5.err
python - <<'PY'
import json
for f in ("c5","c5_stream"):
    try:
        d=json.load(open(f"gpurun_out/r02c/{f}.json"))
        print(f, d["value"], d["ms_per_step"], [(k["codec"],round(k["avg_kernel_ms"],2),round(k["frac"],3)) for k in d["roofline"]["kernels"]], {k:v["mismatches"] for k,v in d["parity"].items()})
    except Exception as e: print(f, "failed", e)
PY
cat gpurun_out/r02c/celt.json
