#!/bin/bash
# Opus front-end on the device: new tests + the stream / CELT suites next to them
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02m
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_opus_gpu.py tests/test_celt_gpu.py tests/test_stream_gpu.py -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/r02m/pytest.log
tail -c 3000 gpurun_out/r02m/pytest.log
