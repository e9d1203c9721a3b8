#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_e2e
for c in ${1:-flac_e2e}; do
  AFG_TRACE=1 timeout 600 python3 tools/bench_codecs.py --codec $c > gpurun_out/r06_e2e/$c.json 2> gpurun_out/r06_e2e/$c.trace
  python3 - "$c" <<'PY'
import json,sys
c=sys.argv[1]
d=json.load(open(f"gpurun_out/r06_e2e/{c}.json"))[c]
print(c, "e2e", round(d["samples_per_s_end_to_end"]/1e9,3), "Gs/s", "ms/call", round(d["seconds"]*1e3,2), "cpu", round(d["cpu_baseline_e2e"]["value"]/1e9,3), "mismatches", d["parity"]["mismatches"])
PY
  tac gpurun_out/r06_e2e/$c.trace | awk "/call set-up/{n++} n==2{exit} {print}" | tac | tail -40
done
