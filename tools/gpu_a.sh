#!/bin/bash
# round-2 first GPU pass: tests, headline bench, C5, launcher
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02a/pytest.log
( timeout 600 python bench.py --steps 10 --warmup 2 2> gpurun_out/r02a/bench.err ) > gpurun_out/r02a/bench.json
( timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --cpu-seconds 6 2> gpurun_out/r02a/c5.err ) > gpurun_out/r02a/c5.json
nproc > gpurun_out/r02a/nproc.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/r02a/nproc.txt 2>&1
tail -c 1500 gpurun_out/r02a/pytest.log; tail -c 600 gpurun_out/r02a/bench.err; head -c 1500 gpurun_out/r02a/bench.json; echo; tail -c 600 gpurun_out/r02a/c5.err; head -c 1200 gpurun_out/r02a/c5.json
