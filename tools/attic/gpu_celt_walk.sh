#!/bin/bash
# round 3: the tolerance-mode CELT walk -- tests, then the dense batch under both numeric modes and a few item sizes
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; tag=${1:-r03_walk}; mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_celt_walk_gpu.py tests/test_celt_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/$tag/pytest.log
cat gpurun_out/$tag/pytest.log
for mode in exact tolerance; do
  AFG_NUMERIC=$mode timeout 600 python tools/bench_codecs.py --codec celt --steps 5 2>&1 | grep '^{' > gpurun_out/$tag/celt_$mode.json
  echo $mode; cat gpurun_out/$tag/celt_$mode.json
done
for seg in 0 64 256 512; do
  AFG_CELT_SEG_RECS=$seg timeout 600 python tools/bench_codecs.py --codec celt --steps 5 2>&1 | grep '^{' > gpurun_out/$tag/celt_seg$seg.json
  echo seg $seg; cat gpurun_out/$tag/celt_seg$seg.json
done
