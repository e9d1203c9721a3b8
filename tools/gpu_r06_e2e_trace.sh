#!/bin/bash
# round 6: where an end-to-end batch call spends its time -- AFG_TRACE laps and a HIP API / copy trace of the FLAC and Vorbis batches
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_e2e
for c in flac_e2e vorbis_e2e mp3_e2e; do
  AFG_TRACE=1 timeout 600 python3 tools/bench_codecs.py --codec $c > gpurun_out/r06_e2e/$c.json 2> gpurun_out/r06_e2e/$c.trace
  python3 - "$c" <<'PY'
import json,sys
c=sys.argv[1]
d=json.load(open(f"gpurun_out/r06_e2e/{c}.json"))[c]
print(c, "e2e", round(d["samples_per_s_end_to_end"]/1e9,3), "Gs/s", "ms/call", round(d["seconds"]*1e3,2), "cpu", round(d["cpu_baseline_e2e"]["value"]/1e9,3))
PY
  # the laps of the last complete call
  tac gpurun_out/r06_e2e/$c.trace | awk '/decode_parsed total/{n++} n==2{exit} {print}' | tac | tail -30
done
cd /tmp
for c in flac_e2e vorbis_e2e; do
  rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r06_e2e/prof_$c" -- python3 "$R/tools/bench_codecs.py" --codec $c > /dev/null 2> "$R/gpurun_out/r06_e2e/prof_$c.err"
  cd "$R"
  for f in $(find gpurun_out/r06_e2e/prof_$c -name "*stats.csv"); do echo "== $f"; head -14 "$f" | cut -c1-200; cp "$f" gpurun_out/r06_e2e/${c}_$(basename $f); done
  rm -rf gpurun_out/r06_e2e/prof_$c
  cd /tmp
done
