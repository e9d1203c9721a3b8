#!/bin/bash
# round-6 evidence: the driver's bench command, the C5 line, a kernel-trace summary of the default bench command, PMC passes per kernel
# (run on the GPU box; the small results are copied into profiles/ afterwards)
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; P=gpurun_out/r06_prof; mkdir -p $P
( timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $P/bench.err ) > $P/bench_line.json
cp gpurun_out/bench_full.json $P/bench_full.json
( timeout 900 python3 bench.py --config c5 --steps 3 --warmup 1 --cpu-seconds 8 --full-record $P/c5_full.json 2> $P/c5.err ) > $P/c5_line.json
cd /tmp
rocprofv3 --kernel-trace --stats -d "$R/$P/stats" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2> "$R/$P/stats.err"
cd "$R"
python3 tools/export_kernel_stats.py $P/stats $P/r06_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-others" > /dev/null
find $P/stats -name "*.db" -delete
bash tools/pmc_collect.sh r06_pmc_mp3_tolerance_kernel mp3_tolerance_kernel bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-full-fetch --no-others > /dev/null
bash tools/pmc_collect.sh r06_pmc_vorbis_walk_kernel vorbis_walk_kernel bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-full-fetch --no-others > /dev/null
bash tools/pmc_collect.sh r06_pmc_vorbis_walk_mc_6ch vorbis_walk_mc_kernel tools/vorbis_shapes.py --only 7 --steps 3 > /dev/null
# FLAC: counters calibrated on the kernel's own access pattern (known bytes), then both populated instantiations together
bash tools/pmc_calib_flac.sh > /dev/null 2>&1
eval $(python3 - <<'PY'
import json
c = json.load(open("gpurun_out/calib_flac/calib.json"))
f = sum(c["FETCH_SIZE_kb_per_launch"]) / len(c["FETCH_SIZE_kb_per_launch"]) * 1024
w = sum(c["WRITE_SIZE_kb_per_launch"]) / len(c["WRITE_SIZE_kb_per_launch"]) * 1024
print(f"export AFG_PMC_FETCH_FACTOR={c['known_read_bytes'] / f:.4f} AFG_PMC_WRITE_FACTOR={c['known_write_bytes'] / w:.4f} AFG_PMC_DISPATCHES_PER_LAUNCH=2")
PY
)
bash tools/pmc_collect.sh r06_pmc_flac_restore1_kernel "flac_restore1_kernel" bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-others > /dev/null
unset AFG_PMC_FETCH_FACTOR AFG_PMC_WRITE_FACTOR AFG_PMC_DISPATCHES_PER_LAUNCH
bash tools/pmc_collect.sh r06_pmc_qoa_decode_kernel qoa_decode_kernel tools/bench_codecs.py --codec qoa --steps 3 > /dev/null
cp gpurun_out/calib_flac/calib.json $P/r06_pmc_calib_flac.json 2>/dev/null
head -12 $P/r06_kernel_stats.csv | cut -c1-160
python3 -c "
import json
d=json.loads(open('$P/bench_line.json').read().strip().splitlines()[-1]); print(len(json.dumps(d)), d['value'], d['ms_per_step'], [ (k['codec'],k['avg_kernel_ms'],k['frac']) for k in d['roofline']['kernels']])
print({k:(v.get('value') or v.get('avg_kernel_ms')) for k,v in d['other_workloads'].items() if isinstance(v,dict)})
c=json.loads(open('$P/c5_line.json').read().strip().splitlines()[-1]); print('c5', c['value'], c['ms_per_step'])
"
for k in mp3_tolerance_kernel vorbis_walk_kernel vorbis_walk_mc_6ch flac_restore1_kernel qoa_decode_kernel; do echo "== $k"; grep -E "hbm_.*bytes|lds_bank|wait_inst_lds|active_inst_valu_over" gpurun_out/r06_pmc_$k/derived.txt; done
