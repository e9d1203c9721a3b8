#!/usr/bin/env python3
"""Print the per-kernel totals of a rocprofv3 run directory (results .db or *_kernel_stats.csv); development helper."""
import glob
import sqlite3
import sys

root = sys.argv[1]
dbs = glob.glob(root + "/**/*_results.db", recursive=True)
if dbs:
    for name, calls, total, avg, pct in sqlite3.connect(dbs[0]).execute("select * from top_kernels limit 12"):
        print(f"{name[:60]:60s} calls {calls:5d}  avg {avg / 1e3:10.3f} ms  {pct:5.1f} %")   # the view reports microseconds
else:
    for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
        print(open(f).read()[:2000])
