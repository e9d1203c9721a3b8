#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
bash $R/tools/pmc_collect.sh r02_pmc_flac_restore_kernel "flac_restore_kernel<8, 12, false, true>" bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline
cat gpurun_out/r02_pmc_flac_restore_kernel/derived.txt
cd /tmp; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02j_c4 -- python3 $R/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cd $R; python tools/top_kernels.py gpurun_out/r02j_c4 | head -20; find gpurun_out/r02j_c4 -name "*.db" -delete
