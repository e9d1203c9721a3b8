#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
bash $R/tools/pmc_collect.sh r02_pmc_mp3_transform_kernel mp3_transform_kernel bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-full-fetch
bash $R/tools/pmc_collect.sh r02_pmc_flac_restore_kernel "flac_restore_kernel<8, 12" bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline
bash $R/tools/pmc_collect.sh r02_pmc_vorbis_wave_kernel vorbis_wave_kernel bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline
cat gpurun_out/r02_pmc_mp3_transform_kernel/derived.txt gpurun_out/r02_pmc_flac_restore_kernel/derived.txt gpurun_out/r02_pmc_vorbis_wave_kernel/derived.txt
