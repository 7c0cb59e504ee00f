#!/usr/bin/env python3
"""Print VGPR/SGPR/scratch/occupancy per kernel of pmx_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "pharmsol_amd/csrc/pmx_kernels.hip"
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-parameter",
                      "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("pmx::(anonymous namespace)::", "").replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"{'kernel':48s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>6s}")
for r in rows:
    print(f"{r['name']:48s} {r.get('vgpr',-1):5d} {r.get('sgpr',-1):5d} {r.get('scratch',-1):8d} {r.get('occ',-1):4d} {r.get('lds',-1):6d}")
