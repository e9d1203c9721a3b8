#!/usr/bin/env python3
"""Print per-kernel mean PMC values from a rocprofv3 results .db or counter_collection csv (development helper)."""
import csv
import glob
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if "pmc_event" in t][0]
    info = [t for t in tabs if "info_pmc" in t][0]
    disp = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, i.name, d.dispatch_id, sum(p.value) from {pmc} p join {info} i on p.pmc_id=i.id "
         f"join {disp} d on p.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by 1,2,3")
    acc = defaultdict(list)
    for k, n, _, v in c.execute(q):
        acc[(k, n)].append(v)
    return acc


def from_csv(path):
    acc = defaultdict(list)
    per = defaultdict(float)
    for r in csv.DictReader(open(path)):
        per[(r["Kernel_Name"], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, n, _), v in per.items():
        acc[(k, n)].append(v)
    return acc


def main():
    for path in sys.argv[1:]:
        files = [path] if not path.endswith("/") else glob.glob(path + "*.db") + glob.glob(path + "*counter_collection.csv")
        for f in files:
            acc = from_db(f) if f.endswith(".db") else from_csv(f)
            for (k, n), v in sorted(acc.items()):
                if "afg" in k or "_kernel" in k:
                    print(f"{k.split('(')[0][-40:]:40s} {n:28s} mean/dispatch {sum(v)/len(v):16.1f}  n={len(v)}")


if __name__ == "__main__":
    main()
