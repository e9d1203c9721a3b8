#!/bin/bash
# round 6, first GPU call: the driver's bench command (is the printed line compact and parseable?), then the counters the
# round-5 review found missing or contaminated: CELT walk, Vorbis walk without the full-fetch leg, the 6-channel Vorbis walk
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_first
( time timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r06_first/bench.err ) > gpurun_out/r06_first/bench.out 2> gpurun_out/r06_first/bench.time
cp gpurun_out/bench_full.json gpurun_out/r06_first/bench_full.json 2>/dev/null
python3 - <<'PY'
import json
t = open("gpurun_out/r06_first/bench.out").read()
lines = [l for l in t.splitlines() if l.strip()]
print("stdout lines:", len(lines), "chars of last:", len(lines[-1]) if lines else 0)
d = json.loads(lines[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
for k in d["roofline"]["kernels"]: print(" ", k["codec"], k["avg_kernel_ms"], k["frac"])
for k, v in d.get("other_workloads", {}).items(): print(" ", k, json.dumps(v)[:300])
PY
tail -3 gpurun_out/r06_first/bench.time
bash tools/pmc_collect.sh r06_pmc_celt_walk_kernel celt_walk_kernel tools/bench_codecs.py --codec celt --steps 3 > /dev/null
bash tools/pmc_collect.sh r06_pmc_vorbis_walk_kernel vorbis_walk_kernel bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-full-fetch --no-others > /dev/null
bash tools/pmc_collect.sh r06_pmc_vorbis_walk_6ch vorbis_walk_kernel tools/vorbis_shapes.py --only 7 --steps 3 > /dev/null
for k in r06_pmc_celt_walk_kernel r06_pmc_vorbis_walk_kernel r06_pmc_vorbis_walk_6ch; do echo "== $k"; cat gpurun_out/$k/derived.txt | head -40; done
