#!/bin/bash
# FLAC launch experiments: tail (file counts), empty instantiations (mask), two streams
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
run() { python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels'][0]
print('$LABEL', 'ms', round(k['avg_kernel_ms'],3), 'ns/ksample', round(k['avg_kernel_ms']*1e6/k['samples_per_launch']*1e3,4), 'mism', d['parity']['flac']['mismatches'])"; }
for f in 1014 1024 1100 1229; do LABEL="files=$f" run --files $f; done
LABEL="mask=0x220" AFG_FLAC_MASK=220 run
LABEL="mask=0x220 two streams" AFG_FLAC_MASK=220 AFG_FLAC_STREAMS=2 run
LABEL="two streams all" AFG_FLAC_STREAMS=2 run
