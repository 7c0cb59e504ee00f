#!/usr/bin/env python3
"""Merge one workload's PMC summary (tools/pmc_summary.py JSON of a tools/pmc_run.sh directory) into
profiles/kernel_counters.json, the file bench.py reads `roofline.traffic` / `roofline.valu_insts` from.
  HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (KB counters; gfx950 correction, MI355X_MICROARCH.md HBM section)
  vector wave-instructions per launch = SQ_INSTS_VALU
usage: tools/kernel_counters.py <workload key: c3|c3_ragged|c2|c4|c5> <pmc_summary.json> <source label>"""
import json, os, sys
key, summary, label = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.load(open(summary))
# the dominant kernel = the one with the most vector instructions (or bytes written)
k = max(d, key=lambda n: (d[n].get("SQ_INSTS_VALU", 0.0), d[n].get("WRITE_SIZE", 0.0)))
c = d[k]
entry = {"kernel": k, "source": label}
if "SQ_INSTS_VALU" in c:
    entry["valu_wave_insts_per_launch"] = c["SQ_INSTS_VALU"]
    entry["salu_insts_per_launch"] = c.get("SQ_INSTS_SALU")
if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
    entry["hbm_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 + 2.0 * c["FETCH_SIZE"] * 1024.0
    entry["WRITE_SIZE_KB"], entry["FETCH_SIZE_KB"] = c["WRITE_SIZE"], c["FETCH_SIZE"]
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "kernel_counters.json")
allc = json.load(open(path)) if os.path.exists(path) else {}
allc[key] = entry
json.dump(allc, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(entry, indent=1))
