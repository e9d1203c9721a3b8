#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
export AFG_LIB_PATH=$R/audio-formats_amd/lib/libafg_abl119.so
bash $R/tools/pmc_collect.sh r02_pmc_vorbis_abl119 tools/bench_codecs.py --codec vorbis --steps 3
