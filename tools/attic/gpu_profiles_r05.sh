#!/bin/bash
# round-5 evidence: bench lines, kernel-trace summary of the default bench command, PMC passes per kernel (run on the GPU box)
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out/r05_prof
( timeout 1500 python bench.py --steps 20 --warmup 3 2> gpurun_out/r05_prof/bench.err ) > gpurun_out/r05_prof/bench.json
( timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --cpu-seconds 8 2> gpurun_out/r05_prof/c5.err ) > gpurun_out/r05_prof/c5.json
cd /tmp
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r05_prof/stats" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2> "$R/gpurun_out/r05_prof/stats.err"
cd "$R"
python tools/export_kernel_stats.py gpurun_out/r05_prof/stats gpurun_out/r05_prof/r05_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-others" > /dev/null
find gpurun_out/r05_prof/stats -name "*.db" -delete
bash tools/pmc_collect.sh r05_pmc_mp3_tolerance_kernel mp3_tolerance_kernel bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-full-fetch > /dev/null
bash tools/pmc_collect.sh r05_pmc_vorbis_walk_kernel vorbis_walk_kernel bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
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
bash tools/pmc_collect.sh r05_pmc_flac_restore1_kernel "flac_restore1_kernel" bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
unset AFG_PMC_FETCH_FACTOR AFG_PMC_WRITE_FACTOR AFG_PMC_DISPATCHES_PER_LAUNCH
bash tools/pmc_collect.sh r05_pmc_qoa_decode_kernel qoa_decode_kernel tools/bench_codecs.py --codec qoa --steps 3 > /dev/null
cp gpurun_out/calib_flac/calib.json gpurun_out/r05_prof/r05_pmc_calib_flac.json 2>/dev/null
head -12 gpurun_out/r05_prof/r05_kernel_stats.csv | cut -c1-160
for k in mp3_tolerance_kernel vorbis_walk_kernel flac_restore1_kernel qoa_decode_kernel; do echo "== $k"; grep -E "hbm_bytes|valu_instructions|lds_instructions|wait_any|active_inst_valu_over|wait_inst_any" gpurun_out/r05_pmc_$k/derived.txt; done
