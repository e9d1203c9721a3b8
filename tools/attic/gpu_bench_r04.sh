#!/bin/bash
# tools/gpu_bench_r04.sh: the full default bench line (as the driver runs it) + the 2-rank launcher test
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"
mkdir -p gpurun_out
( time python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/r04_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
for k in d['roofline']['kernels']: print(k['codec'], round(k['avg_kernel_ms'],3), round(k['frac'],4))
print('parity', {k:(v['mismatches'], v.get('rms_error')) for k,v in d['parity'].items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
ow=d.get('other_workloads',{})
for k,v in ow.items():
    if isinstance(v,dict):
        keep={x:v[x] for x in ('value','ms_per_step','avg_kernel_ms','frac','samples_per_s_end_to_end','seconds','error','vs_cpu_baseline_e2e','res16_frame_fraction') if x in v}
        if 'cpu_baseline_e2e' in v: keep['cpu_e2e']=v['cpu_baseline_e2e']['value']
        if k=='device_inclusive': keep={c:round(x['samples_per_s_device_inclusive']/1e9,2) for c,x in v.items()}
        print(k, keep)
print('wall others', ow.get('codecs_wall_s'))
PY
timeout 1200 python -m pytest tests/test_multidevice_gpu.py -m gpu -x -q -k "bench_launches" 2>&1 | tail -3
