#!/bin/bash
# kernel-trace summary of the headline step alone (no full-fetch legs, no side-by-side step): the averages bench.py's per-kernel event times must agree with
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; P=gpurun_out/r06_prof; mkdir -p $P
cd /tmp
rocprofv3 --kernel-trace --stats -d "$R/$P/stats2" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-others --no-full-fetch > "$R/$P/stats2_line.json" 2> "$R/$P/stats2.err"
cd "$R"
python3 tools/export_kernel_stats.py $P/stats2 $P/r06_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-others --no-full-fetch" > /dev/null
find $P/stats2 -name "*.db" -delete
head -7 $P/r06_kernel_stats.csv | cut -c1-220
python3 -c "
import json
d=json.loads(open('$P/stats2_line.json').read().strip().splitlines()[-1]); print([(k['codec'],k['avg_kernel_ms']) for k in d['roofline']['kernels']])"
python3 tools/celt_decompose.py --steps 5
