#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; mkdir -p gpurun_out/r02e
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_celt_gpu.py tests/test_multidevice_gpu.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r02e/pytest.log
( timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/r02e/c5.err ) > gpurun_out/r02e/c5.json
( timeout 600 python tools/bench_codecs.py --codec celt --steps 5 --warmup 2 2>&1 | tail -3 ) > gpurun_out/r02e/celt.json
cd /tmp
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r02e/c5prof" -- python3 "$R/bench.py" --config c5 --c5-files 8192 --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/r02e/c5p.json" 2> "$R/gpurun_out/r02e/c5p.err"
cd "$R"
cat gpurun_out/r02e/pytest.log; tail -c 400 gpurun_out/r02e/c5.err
python - <<'PY'
import json
for f in ("c5",):
    try:
        d=json.load(open(f"gpurun_out/r02e/{f}.json"))
        print(f, d["value"], d["ms_per_step"], [(k["codec"],round(k["avg_kernel_ms"],2),round(k["frac"],3)) for k in d["roofline"]["kernels"]], {k:v["mismatches"] for k,v in d["parity"].items()})
    except Exception as e: print(f, "failed", e)
PY
cat gpurun_out/r02e/celt.json; python tools/top_kernels.py gpurun_out/r02e/c5prof | head -8
