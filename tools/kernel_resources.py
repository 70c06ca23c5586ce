#!/usr/bin/env python3
"""Print VGPR/AGPR/spill/LDS/occupancy per kernel of a .hip source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s*(Function Name): (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", "-p", m.group(2)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for c in rows:
    n = c["name"]
    n = re.sub(r"\(.*", "", n)
    print(f"V={c.get('VGPRs', -1):3d} A={c.get('AGPRs', -1):3d} spillV={c.get('VGPRs Spill', -1)} scratch={c.get('ScratchSize', -1)} "
          f"occ={c.get('Occupancy', -1)} lds={c.get('LDS Size', -1)}  {n[:150]}")
if r.returncode:
    print(r.stderr[-3000:])
