#!/bin/bash
# full GPU check of the round: every -m gpu test, the smoke entry, the headline bench and the C5 bench
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; tag=${1:-r02_full}; mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/$tag/pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/$tag/smoke.log
( timeout 900 python bench.py --steps 10 --warmup 2 2> gpurun_out/$tag/bench.err ) > gpurun_out/$tag/bench.json
( timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --cpu-seconds 6 2> gpurun_out/$tag/c5.err ) > gpurun_out/$tag/c5.json
cat gpurun_out/$tag/pytest.log gpurun_out/$tag/smoke.log
python - <<PY
import json
for f in ("bench","c5"):
    try:
        d=json.load(open("gpurun_out/$tag/%s.json" % f))
        print(f, "%.4g" % d["value"], round(d["ms_per_step"],2), [(k["codec"],round(k["avg_kernel_ms"],2),round(k["frac"],3)) for k in d["roofline"]["kernels"]], {k:v["mismatches"] for k,v in d["parity"].items()}, d["cpu_baseline"] and "%.3g" % d["cpu_baseline"]["value"])
    except Exception as e: print(f, "failed", e)
PY
