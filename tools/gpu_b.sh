#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
( AFG_CELT_PATH=split timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/r02b/c5_split.err ) > gpurun_out/r02b/c5_split.json
cd /tmp
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02b/stats" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/r02b/bench_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r02b/bench_prof.err"
cd "$GRAFT_REPO_ROOT"
find gpurun_out/r02b/stats -name "*kernel_stats*" | head; 
f=$(find gpurun_out/r02b/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
head -c 3000 gpurun_out/r02b/c5_split.json; tail -c 500 gpurun_out/r02b/c5_split.err
