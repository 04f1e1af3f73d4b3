#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch
(largest dispatches only, i.e. grid size == max for that kernel)."""
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for p in glob.glob(f + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0]
            rows[(k, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
best = {}
for (k, gs) in rows:
    if k not in best or gs > best[k]:
        best[k] = gs
for k, gs in sorted(best.items()):
    d = rows[(k, gs)]
    print(k, "grid", gs, "n", len(next(iter(d.values()))))
    for c, v in sorted(d.items()):
        print("    %-28s %.4g" % (c, sum(v) / len(v)))
