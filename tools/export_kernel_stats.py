#!/usr/bin/env python3
"""tools/export_kernel_stats.py <rocprofv3 output dir> <out.csv> [command text]
Writes the per-kernel summary of a `rocprofv3 --kernel-trace --stats` run (results .db) as a small CSV for profiles/."""
import glob
import sqlite3
import sys

root, out = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
db = glob.glob(root + "/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                      f"from {disp} d join {sym} s on d.kernel_id = s.id group by 1 order by 3 desc"))
total = sum(r[2] for r in rows) or 1
with open(out, "w") as fh:
    fh.write(f"# {cmd}   (durations in ns; kernels of the product library first, then the synthetic-input generators)\n")
    fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for name, calls, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-110:].replace(",", ";")
        fh.write(f"{short},{calls},{tot},{avg:.1f},{100.0 * tot / total:.2f},{mn},{mx}\n")
print(open(out).read()[:1500])
