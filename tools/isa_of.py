#!/usr/bin/env python3
"""Print the gfx950 ISA of ONE kernel of pmx_kernels.hip (comment-free), e.g.
   tools/isa_of.py 'pmx_analytical_stepsILi4ELb0E' [/tmp/pmx_kernels.s]   (run tools/isa_count.sh first to produce the .s)"""
import re, sys
pat, path = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "/tmp/pmx_kernels.s")
txt = open(path).read()
m = re.search(r"^(\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M)
name = m.group(1)
for l in m.group(2).split("\n"):
    if l.strip() and not l.strip().startswith(";") and not l.strip().startswith(".p2align"):
        print(l.split(";")[0].rstrip() if not l.startswith(".LBB") else l)
for key in ("num_vgpr", "numbered_sgpr", "private_seg_size"):
    mm = re.search(r"\.set " + re.escape(name) + r"\." + key + r", (\d+)", txt)
    print(";", key, mm.group(1) if mm else "?")
