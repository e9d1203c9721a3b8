#!/bin/bash
# round 6, final numbers: the driver's bench command (compact line + full record) and the C5 line
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; P=gpurun_out/r06_final; mkdir -p $P
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $P/bench.err ) > $P/bench_line.json 2> $P/bench.time
cp gpurun_out/bench_full.json $P/bench_full.json
( timeout 900 python3 bench.py --config c5 --steps 3 --warmup 1 --cpu-seconds 8 --full-record $P/c5_full.json 2> $P/c5.err ) > $P/c5_line.json
tail -3 $P/bench.time
python3 -c "
import json
t=open('$P/bench_line.json').read().strip().splitlines(); print('stdout lines', len(t), 'chars', len(t[-1]))
d=json.loads(t[-1]); print(d['value'], d['ms_per_step'], [(k['codec'],k['avg_kernel_ms'],k['frac']) for k in d['roofline']['kernels']], 'cpu', d['cpu_baseline']['value'])
for k,v in d['other_workloads'].items(): print(' ', k, json.dumps(v)[:260])
c=json.loads(open('$P/c5_line.json').read().strip().splitlines()[-1]); print('c5', c['value'], c['ms_per_step'])
"
