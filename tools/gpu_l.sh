#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd $R
timeout 900 python -m pytest tests/test_mp3_requant_gpu.py tests/test_stream_gpu.py tests/test_multidevice_gpu.py -x -q 2>&1 | tail -5
for m in q f; do
  if [ $m = f ]; then export AFG_MP3_FLOAT_UPLOAD=1; else unset AFG_MP3_FLOAT_UPLOAD; fi
  echo "== upload $m"; AFG_TRACE=1 python tools/bench_codecs.py --codec mp3_e2e --e2e-files 2048 2>&1 | grep -E "mp3 parse|pipeline drain|samples_per_s" | tail -4 | cut -c1-300
done
timeout 300 python tools/fuzz/fuzz_batch.py 2>&1 | tail -3
