#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; R=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/r02gap" -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-full-fetch > /dev/null 2>&1
cd "$R"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02gap/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("mp3_transform", "vorbis_wave", "flac_restore"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 3 steps
ev = [(r["Kernel_Name"][20:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
big = [e for e in ev if e[2] - e[1] > 1_000_000]
for a, b in zip(big[-12:-1], big[-11:]):
    print(a[0], "dur %.3f ms" % ((a[2]-a[1])/1e6), "gap to next start %.3f ms" % ((b[1]-a[2])/1e6))
PY
find gpurun_out/r02gap -name "*.csv" -size +1M -delete
