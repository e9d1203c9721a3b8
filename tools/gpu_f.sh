#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r02f
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r02f/celt_only" -- python3 "$R/bench.py" --config c5 --c5-files 8192 --only celt --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/r02f/c.json" 2> "$R/gpurun_out/r02f/c.err"
cd $R; python tools/top_kernels.py gpurun_out/r02f/celt_only | head -5; tail -3 gpurun_out/r02f/c.err
