#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: duration of the LAST K dispatches of the dominant pmx kernel (= the K timed passes
of bench.py, which come last), next to the all-dispatch average the --stats summary prints (that one also holds the
placement-search, spin-up and warm-up passes).
usage: tools/trace_last_k.py <kernel_trace.csv> <K> [bench.json]  -> JSON on stdout"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2])
byk = {}
for r in rows:
    n = r.get("Kernel_Name", "")
    if "pmx" not in n:
        continue
    byk.setdefault(n, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
name = max(byk, key=lambda n: sum(e - s for s, e in byk[n]))
d = sorted(byk[name])
dur = [(e - s) / 1e6 for s, e in d]
last = dur[-K:]
out = {"kernel": name[:100], "dispatches": len(dur), "all_mean_ms": sum(dur) / len(dur), "all_min_ms": min(dur), "all_max_ms": max(dur),
       "last_k": K, "last_k_mean_ms": sum(last) / len(last), "last_k_min_ms": min(last), "last_k_max_ms": max(last)}
if len(sys.argv) > 3:
    b = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    out["bench_kernel_ms"] = b["roofline"]["kernel_ms"]
    out["bench_ms_per_step"] = b["ms_per_step"]
    out["bench_frac"] = b["roofline"]["frac"]
    out["agreement"] = out["last_k_mean_ms"] / b["roofline"]["kernel_ms"]
print(json.dumps(out, indent=1))
