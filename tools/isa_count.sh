#!/bin/bash
# Count VALU instructions per kernel of pmx_kernels.hip (whole function; compare before/after a change).
# usage: tools/isa_count.sh [out.s]   (writes the gfx950 assembly there, default /tmp/pmx_kernels.s)
out=${1:-/tmp/pmx_kernels.s}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-parameter -Iinclude --cuda-device-only -S \
  pharmsol_amd/csrc/pmx_kernels.hip -o "$out" || exit 1
python3 - "$out" <<'PY'
import re, subprocess, sys
name, rows = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\S+):", line)
    if m:
        name = m.group(1); rows[name] = [0, 0, 0, 0]; continue
    if name is None: continue
    t = line.strip()
    if t.startswith("s_endpgm"): name = None; continue
    if t.startswith("v_mov_b64") or t.startswith("v_mov_b32"): rows[name][1] += 1
    if re.match(r"v_(fma|fmac|mul|add)_f64", t): rows[name][2] += 1
    if t.startswith("v_"): rows[name][0] += 1
    if t.startswith("s_"): rows[name][3] += 1
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':48s} {'VALU':>6s} {'v_mov':>6s} {'f64 arith':>9s} {'SALU':>6s}")
for full, (k, v) in zip(names, rows.items()):
    mm = re.search(r"(pmx_\w+<[^>]*>)", full)
    if mm: print(f"{mm.group(1):48s} {v[0]:6d} {v[1]:6d} {v[2]:9d} {v[3]:6d}")
PY
