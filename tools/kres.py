#!/usr/bin/env python3
"""Per-kernel resource usage of the library (VGPRs, AGPRs, scratch, LDS, occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage remarks:  python tools/kres.py [source] [extra hipcc flags]"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1] if len(sys.argv) > 1 else "super_resolution_amd/csrc/gpsr_api.hip"
with tempfile.NamedTemporaryFile(suffix=".so") as tmp:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
                          "-Rpass-analysis=kernel-resource-usage", "-o", tmp.name, src] + sys.argv[2:],
                         capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0]] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = name.split("(")[0]
    print("%-58s vgpr %4d agpr %4d scratch %5d occ %2d lds %6d" % (name, v.get("VGPRs", -1), v.get("AGPRs", -1),
          v.get("ScratchSize", -1), v.get("Occupancy", -1), v.get("LDS", -1)))
