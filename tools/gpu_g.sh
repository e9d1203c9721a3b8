#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r02g
rocprofv3 --list-avail > $R/gpurun_out/r02g/avail.txt 2>&1
grep -c "" $R/gpurun_out/r02g/avail.txt
grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|LDS[A-Za-z_0-9]*\|FETCH_SIZE\|WRITE_SIZE" $R/gpurun_out/r02g/avail.txt | sort -u | tr '\n' ' ' | head -c 6000
