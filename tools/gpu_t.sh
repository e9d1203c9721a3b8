#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/pmc_calib_flac.sh > /dev/null 2>&1
eval $(python3 - <<'PY'
import json
c = json.load(open("gpurun_out/calib_flac/calib.json"))
f = sum(c["FETCH_SIZE_kb_per_launch"]) / len(c["FETCH_SIZE_kb_per_launch"]) * 1024
w = sum(c["WRITE_SIZE_kb_per_launch"]) / len(c["WRITE_SIZE_kb_per_launch"]) * 1024
print(f"export AFG_PMC_FETCH_FACTOR={c['known_read_bytes'] / f:.4f} AFG_PMC_WRITE_FACTOR={c['known_write_bytes'] / w:.4f} AFG_PMC_DISPATCHES_PER_LAUNCH=2")
PY
)
env | grep AFG_PMC
bash tools/pmc_collect.sh r02_pmc_flac_restore_kernel "flac_restore_kernel" bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
cat gpurun_out/r02_pmc_flac_restore_kernel/derived.txt | grep -E "hbm|factor|fetch|write|dispatch"
