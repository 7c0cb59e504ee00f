#!/usr/bin/env python3
"""Print VGPR/SGPR/scratch/occupancy per kernel of pmx_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py [substring-filter]"""
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
src = "pharmsol_amd/csrc/pmx_kernels.hip"
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-parameter", "-Iinclude",
                      "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
pats = (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        mm = re.search(r"(pmx_\w+<[^>]*>)", name)
        cur = {"name": mm.group(1) if mm else name[:60]}
        rows.append(cur)
        continue
    for key, pat in pats:
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"{'kernel':44s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>6s}")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name']:44s} {r.get('vgpr',-1):5d} {r.get('sgpr',-1):5d} {r.get('scratch',-1):8d} {r.get('occ',-1):4d} {r.get('lds',-1):6d}")
