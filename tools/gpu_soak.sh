#!/bin/bash
# the -m gpu suite several times in a row (fresh processes): flaky failures show here, not at round end
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out/soak
n=${1:-6}
for i in $(seq 1 $n); do
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/soak/run$i.log
  echo "run $i: $(tail -1 gpurun_out/soak/run$i.log)"
done
