#!/bin/bash
# Layer I / II on the device + the MP3 suites next to it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02n
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_mp3_layer12_gpu.py tests/test_mp3_gpu.py tests/test_mp3_requant_gpu.py tests/test_stream_gpu.py -m gpu -x -q 2>&1 | tail -40 ) > gpurun_out/r02n/pytest.log
tail -c 3000 gpurun_out/r02n/pytest.log
