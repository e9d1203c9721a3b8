#!/bin/bash
# the grouped batch pipeline: its test, then the end-to-end legs over group counts and helper-thread counts
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_e2e
( timeout 900 python3 -m pytest tests/test_default_mode_gpu.py tests/test_stream_gpu.py tests/test_multidevice_gpu.py -m gpu -x -q 2>&1 | tail -5 )
for c in ${CODECS:-flac_e2e vorbis_e2e mp3_e2e}; do
  for cfg in ${@:-1:0 4:0 6:0 8:0 6:16 6:24 6:32 8:24}; do
    g=${cfg%%:*}; t=${cfg##*:}
    AFG_BATCH_GROUPS=$g timeout 600 python3 tools/bench_codecs.py --codec $c --e2e-threads $t > gpurun_out/r06_e2e/${c}_g${g}_t$t.json 2> /dev/null
    python3 - "$c" "$g" "$t" <<'PY'
import json,sys
c,g,t=sys.argv[1:4]
d=json.load(open(f"gpurun_out/r06_e2e/{c}_g{g}_t{t}.json"))[c]
print(c, "groups", g, "threads", t, "e2e", round(d["samples_per_s_end_to_end"]/1e9,3), "Gs/s", "ms/call", round(d["seconds"]*1e3,2), "cpu-s/call", round(d.get("host_cpu_seconds_per_call",0),3), "cpus busy", round(d.get("host_cpus_busy",0),1), "mismatches", d["parity"]["mismatches"])
PY
  done
done
