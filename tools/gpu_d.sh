#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; mkdir -p gpurun_out/r02d
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_transcode_gpu.py tests/test_wav_writer.py -x -q 2>&1 | tail -8 ) > gpurun_out/r02d/pytest.log
cd /tmp
for path in split stream; do
  AFG_CELT_PATH=$path rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r02d/c5_$path" -- python3 "$R/bench.py" --config c5 --c5-files 8192 --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/r02d/c5_$path.json" 2> "$R/gpurun_out/r02d/c5_$path.err"
done
cd "$R"
cat gpurun_out/r02d/pytest.log
for path in split stream; do echo "== $path"; python tools/top_kernels.py gpurun_out/r02d/c5_$path | head -9; done
