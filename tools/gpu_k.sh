#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
bash $R/tools/pmc_collect.sh r02_pmc_flac_restore_kernel "flac_restore_kernel<8, 12, false, true" bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
bash $R/tools/pmc_collect.sh r02_pmc_celt_stream_kernel "celt_stream_kernel" tools/bench_codecs.py --codec celt --steps 3 > /dev/null
bash $R/tools/pmc_collect.sh r02_pmc_qoa_decode_kernel "qoa_decode_kernel" tools/bench_codecs.py --codec qoa --steps 3 > /dev/null
for t in flac_restore_kernel celt_stream_kernel qoa_decode_kernel; do echo "== $t"; cat gpurun_out/r02_pmc_$t/derived.txt; done 2>&1 | grep -v "^ \"s[aq]_\|smem" 
