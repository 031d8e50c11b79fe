"""Print registers / spills / LDS / occupancy of every kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py pw6.hip [name filter]"""
import os
import re
import subprocess
import sys

here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "x3d-multigrid_amd", "csrc")
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-fno-slp-vectorize"] if src.startswith("dw") else []
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(here, src), "-o", "/tmp/_kr.o"] + extra
out = subprocess.run(cmd, capture_output=True, text=True, stdin=subprocess.DEVNULL, timeout=1500).stderr
rows, cur = [], None
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
    for k, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                   ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                   ("spill", r"VGPRs Spill: (\d+)")):
        m = re.search(pat, l)
        if m and cur is not None:
            cur[k] = int(m.group(1))
if not rows:
    sys.exit("no kernels reported -- compile errors?\n" + out[-3000:])
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).split("(")[0]
    if flt in n:
        print("%-70s vgpr %3d agpr %3d spill %3d scratch %4d lds %6d occ %d" % (
            n[:70], r.get("vgpr", -1), r.get("agpr", -1), r.get("spill", -1), r.get("scratch", -1), r.get("lds", -1), r.get("occ", -1)))
