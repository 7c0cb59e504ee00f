#!/usr/bin/env python3
"""ISA guard for the write-bound kernels: the prediction stores of the classed kernels must stay fire-and-forget.

On gfx9-class hardware loads and stores share ONE in-order counter (vmcnt), so a single `s_waitcnt vmcnt(..)` that the
compiler places in the emit path - e.g. because a rarely taken branch left a vector load pending at a join - makes every
observation step wait for all earlier prediction stores.  That cost the C3 headline 15 % once this round (0.83 -> 0.98 ms)
without changing one line of the hot loop's source.  This script compiles pmx_kernels.hip to assembly and checks, for
every exact and loose prediction instantiation of pmx_analytical_classed, that no basic block holding a 16-byte
prediction store also holds a vmcnt wait.
usage: tools/isa_guard.py [path/to/pmx_kernels.s]   (compiles to pharmsol_amd/csrc/build/pmx_kernels.s when absent/stale)"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pharmsol_amd", "csrc", "pmx_kernels.hip")
OUT = os.path.join(ROOT, "pharmsol_amd", "csrc", "build", "pmx_kernels.s")


def assembly(path=None):
    if path:
        return open(path).read()
    deps = [SRC] + [os.path.join(ROOT, "pharmsol_amd", "csrc", h) for h in
                    ("pmx_structures.hpp", "pmx_device.hpp", "pmx_devtypes.hpp", "pmx_ode.hpp", "pmx_kernels.hpp")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-parameter",
                        "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", SRC, "-o", OUT], check=True,
                       stderr=subprocess.DEVNULL)
    return open(OUT).read()


def check(text):
    """-> (n kernels checked, [violations])"""
    bad, n = [], 0
    name, block, blocks = None, None, {}
    def finish():
        nonlocal n
        if name is None:
            return
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
        m = re.search(r"pmx_analytical_classed<(\d+), (\w+), (\w+), (\w+), (\w+), (\w+)>", dem)
        if not m or m.group(2) != "false" or m.group(4) != "false" or m.group(6) != "false":
            return  # prediction mode, no lag class, no covariate rebuild: the write-bound instantiations
        n += 1
        for b, (waits, stores) in blocks.items():
            if waits and stores:
                bad.append(f"{m.group(0)}: block {b} holds {stores} prediction store(s) and {waits} vmcnt wait(s)")
    for line in text.split("\n"):
        m = re.match(r"^(_ZN3pmx\S*pmx_analytical_classed\S*):", line)
        if m:
            finish()
            name, block, blocks = m.group(1), "entry", {}
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            finish()
            name = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            block = m.group(1)
        d = blocks.setdefault(block, [0, 0])
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            d[0] += 1
        if t.startswith("global_store_dwordx4"):
            d[1] += 1
    finish()
    return n, bad


if __name__ == "__main__":
    n, bad = check(assembly(sys.argv[1] if len(sys.argv) > 1 else None))
    print(f"{n} write-bound classed instantiations checked")
    for b in bad:
        print("VIOLATION", b)
    sys.exit(1 if bad or n == 0 else 0)
