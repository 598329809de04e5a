#!/usr/bin/env python3
"""Print a per-kernel resource table (VGPR/AGPR/SGPR/spills/scratch/LDS/occupancy) for one .hip file."""
import re, subprocess, sys, os
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "-c", src, "-o", "/tmp/_ru.o", "--offload-arch=gfx950", "-O3", "-fPIC",
       "-std=c++17", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
r = subprocess.run(cmd, capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "").replace("void ", "")
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
keys = ["VGPRs", "AGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]
print("%-60s %5s %5s %5s %7s %7s %8s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "sSpill", "vSpill", "scratch", "LDS", "occ"))
for k, d in rows.items():
    print("%-60s %5s %5s %5s %7s %7s %8s %7s %4s" % ((k[:60],) + tuple(d.get(x, "?") for x in keys)))
if r.returncode: print(r.stderr[-3000:])
