#!/bin/bash
# tools/pmc_collect.sh <tag> <kernel-name-substring> <program args...>   (runs ON the GPU box, from the repo root)
# Separate rocprofv3 --pmc passes (counters only, never combined with tracing) over `python3 <args>`, one results .db
# per pass under gpurun_out/<tag>/pass<k>/.  tools/pmc_summary.py condenses them into profiles/.
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
tag=$1; shift; needle=$1; shift; prog=$1; shift
export TMPDIR=/tmp
cd /tmp
k=0
# the counter sets, one rocprofv3 pass each (AFG_PMC_SETS="FETCH_SIZE;WRITE_SIZE" restricts them: bench.py --measure-traffic)
if [ -n "$AFG_PMC_SETS" ]; then IFS=';' read -r -a sets <<< "$AFG_PMC_SETS"; else sets=(
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
  "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"); fi
for set in "${sets[@]}"; do
  k=$((k+1))
  mkdir -p "$R/gpurun_out/$tag/pass$k"
  echo "$set" > "$R/gpurun_out/$tag/pass$k/counters.txt"
  rocprofv3 --pmc $set -d "$R/gpurun_out/$tag/pass$k" -- python3 "$R/$prog" "$@" > "$R/gpurun_out/$tag/pass$k/stdout.log" 2> "$R/gpurun_out/$tag/pass$k/stderr.log"
  echo "pass $k ($set): rc $?"
done
cd "$R"
# condense here: the raw .db files (12 MB per pass) would not fit the 64 MiB that travels back
python3 tools/pmc_summary.py "$tag" "$needle" "gpurun_out/$tag/$tag.json" > "gpurun_out/$tag/derived.txt"
find "gpurun_out/$tag" -name "*.db" -delete
