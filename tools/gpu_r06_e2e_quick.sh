#!/bin/bash
# the end-to-end legs without tracing: samples/s, ms per call, the CPU baseline beside it
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_e2e
for c in ${@:-flac_e2e vorbis_e2e mp3_e2e}; do
  timeout 600 python3 tools/bench_codecs.py --codec $c > gpurun_out/r06_e2e/$c.json 2> gpurun_out/r06_e2e/$c.err
  python3 - "$c" <<'PY'
import json,sys
c=sys.argv[1]
d=json.load(open(f"gpurun_out/r06_e2e/{c}.json"))[c]
print(c, "e2e", round(d["samples_per_s_end_to_end"]/1e9,3), "Gs/s", "ms/call", round(d["seconds"]*1e3,2), "windows", [round(x*1e3,1) for x in d["seconds_per_call_windows"]], "cpu", round(d["cpu_baseline_e2e"]["value"]/1e9,3), "mismatches", d["parity"]["mismatches"])
PY
done
