#!/bin/bash
# end-to-end legs against the number of helper threads: on a box with a CPU quota the host parse's CPU seconds bound the rate
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_e2e
for c in ${CODECS:-flac_e2e vorbis_e2e}; do
  for t in ${@:-16 24 32 64 0}; do
    timeout 600 python3 tools/bench_codecs.py --codec $c --e2e-threads $t > gpurun_out/r06_e2e/${c}_t$t.json 2> /dev/null
    python3 - "$c" "$t" <<'PY'
import json,sys
c,t=sys.argv[1],sys.argv[2]
d=json.load(open(f"gpurun_out/r06_e2e/{c}_t{t}.json"))[c]
print(c, "threads", t, "e2e", round(d["samples_per_s_end_to_end"]/1e9,3), "Gs/s", "ms/call", round(d["seconds"]*1e3,2), "cpu-s/call", round(d.get("host_cpu_seconds_per_call",0),3), "cpus busy", round(d.get("host_cpus_busy",0),1))
PY
  done
done
