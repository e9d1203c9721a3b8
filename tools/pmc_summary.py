#!/usr/bin/env python3
"""tools/pmc_summary.py <tag> <kernel-name-substring> [out.json]

Condense the rocprofv3 --pmc passes tools/pmc_collect.sh left under gpurun_out/<tag>/pass*/ into one small JSON under
profiles/: per-dispatch means of every counter for the kernels whose name contains the substring, the derived ratios the
DESIGN text quotes (instructions per wavefront, issue utilisation, LDS conflict share, HBM bytes), and the command."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_db(path, needle):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]
    info = [t for t in tabs if "info_pmc" in t][0]
    disp = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, i.name, d.dispatch_id, sum(p.value) from {pmc} p join {info} i on p.pmc_id=i.id "
         f"join {disp} d on p.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by 1,2,3")
    acc = defaultdict(list)
    names = set()
    for k, n, _, v in c.execute(q):
        names.add(k)
        if needle in k:
            acc[n].append(v)
    if not acc:
        sys.stderr.write(f"no kernel name contains {needle!r}; kernels seen: {sorted(x[:90] for x in names)}\n")
    return acc


def main():
    tag, needle = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", f"{tag}.json")
    counters, dispatches, cmd = {}, {}, None
    for p in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "pass*"))):
        dbs = glob.glob(os.path.join(p, "**", "*_results.db"), recursive=True)
        if not dbs:
            continue
        for name, vals in read_db(dbs[0], needle).items():
            vals = vals[1:] if len(vals) > 2 else vals          # drop the first (cold) dispatch
            counters[name] = sum(vals) / len(vals)
            dispatches[name] = len(vals)
    c = counters
    derived = {}
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        w = c["SQ_WAVES"]
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
            if k in c:
                derived[k.replace("SQ_INSTS_", "").lower() + "_instructions_per_wavefront"] = c[k] / w
    if c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if k in c:
                derived[k.lower() + "_over_wave_cycles"] = c[k] / wc
    if c.get("SQ_LDS_IDX_ACTIVE"):
        derived["lds_bank_conflict_cycles_over_lds_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 the
        # derived FETCH_SIZE counts 64-byte requests as 32 bytes: doubled here
        # A kernel whose access width the guide does not list brings its own factors, measured on its access pattern with
        # known byte counts (FLAC: tools/pmc_calib_flac.sh -> profiles/r02_pmc_calib_flac.json), and says how many of the
        # averaged dispatches make one launch of the library entry (FLAC: the two populated instantiations).
        ff = float(os.environ.get("AFG_PMC_FETCH_FACTOR", "2.0"))
        wf = float(os.environ.get("AFG_PMC_WRITE_FACTOR", "1.0"))
        per_launch = float(os.environ.get("AFG_PMC_DISPATCHES_PER_LAUNCH", "1"))
        f = c.get("FETCH_SIZE", 0.0) * 1024 * ff * per_launch
        wv = c.get("WRITE_SIZE", 0.0) * 1024 * wf * per_launch
        derived["hbm_fetch_bytes_per_launch"] = f
        derived["hbm_write_bytes_per_launch"] = wv
        derived["hbm_bytes_per_launch"] = f + wv
        derived["hbm_counter_factors"] = {"fetch": ff, "write": wf, "dispatches_per_launch": per_launch}
    res = {"kernel_name_contains": needle,
           "how": "tools/pmc_collect.sh: separate `rocprofv3 --pmc <set> -- python3 <program>` passes (counters only), means over the "
                  "dispatches after the first; SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md)",
           "counters_mean_per_dispatch": counters, "dispatches_averaged": dispatches, "derived": derived}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
