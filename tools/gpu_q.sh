#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r02q; export TMPDIR=/tmp
for v in 1 ""; do
  echo "AFG_FLAC_HOST_RES32=$v"
  AFG_FLAC_HOST_RES32=$v AFG_TRACE=1 timeout 600 python tools/bench_codecs.py --codec flac_e2e --steps 5 --warmup 2 2> gpurun_out/r02q/fe2e_$v.err | tail -1 | cut -c1-400
  tail -14 gpurun_out/r02q/fe2e_$v.err
done
