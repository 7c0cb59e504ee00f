#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/pmc_run.sh output): mean counter value per dispatch of the pmx kernel."""
import csv, glob, sys, collections, json
d = sys.argv[1]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(f"{d}/pass*/pmc_counter_collection.csv")):
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "pmx" not in r["Kernel_Name"]:
            continue
        per_dispatch[(r["Dispatch_Id"], r["Counter_Name"], r["Kernel_Name"][:60])] += float(r["Counter_Value"])
    for (disp, name, k), v in per_dispatch.items():
        acc[(k, name)].append(v)
out = {}
for (k, name), vals in sorted(acc.items()):
    out.setdefault(k, {})[name] = sum(vals) / len(vals)
for k, c in out.items():
    print(k)
    for name, v in c.items():
        print(f"   {name:28s} {v:18.1f}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
