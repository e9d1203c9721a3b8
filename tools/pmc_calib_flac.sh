#!/bin/bash
# FETCH_SIZE / WRITE_SIZE on the FLAC restore kernel's access pattern with known byte counts (tools/ubench_flacpattern.hip calib):
# the calibration MI355X_MICROARCH.md asks for before trusting the counters on an access width it does not list.
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$R/tools/ubench_flacpattern.bin" "$R/tools/ubench_flacpattern.hip" 2>/dev/null
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p "$R/gpurun_out/calib_flac/$c"
  rocprofv3 --pmc $c -d "$R/gpurun_out/calib_flac/$c" -- "$R/tools/ubench_flacpattern.bin" calib > "$R/gpurun_out/calib_flac/$c.log" 2>&1
done
cd "$R"
python3 - <<'PY'
import glob, sqlite3, json
out = {"pattern": "k16<32,64,nt>: 32 frames x 2 channels, 128-byte int16 row reads, 512-byte nontemporal interleaved int32 row writes, C4 size",
       "known_read_bytes": 1323008 * 8192 * 2, "known_write_bytes": 1323008 * 8192 * 4}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob(f"gpurun_out/calib_flac/{c}/**/*_results.db", recursive=True)[0]
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]; disp = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(con.execute(f"select s.kernel_name, d.dispatch_id, sum(p.value) from {pmc} p join {disp} d on p.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by 1,2"))
    vals = [v for k, _, v in rows if "k16" in k]
    out[c + "_kb_per_launch"] = vals
print(json.dumps(out))
json.dump(out, open("gpurun_out/calib_flac/calib.json", "w"), indent=1)
PY
find gpurun_out/calib_flac -name "*.db" -delete
cat gpurun_out/calib_flac/FETCH_SIZE.log | tail -3
