#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_e2e
for x in ${@:-8}; do
  AFG_E2E_EXP=$x AFG_TRACE=1 timeout 600 python3 tools/bench_codecs.py --codec flac_e2e --e2e-distinct 64 > gpurun_out/r06_e2e/exp$x.json 2> gpurun_out/r06_e2e/exp$x.trace
  echo "== exp $x"; tac gpurun_out/r06_e2e/exp$x.trace | awk '/decode_parsed total/{n++} n==2{exit} {print}' | tac | grep -E "flac stage|chunk 3|chunk 6|pass 1b|probe"
done
