#!/usr/bin/env python3
"""usage: tools/kres.py file.hip [filter] -- per-kernel VGPR / spill / scratch / occupancy table of one translation unit (gfx950)."""
import os, re, subprocess, sys
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lighthand_amd", "csrc")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + sys.argv[3:], cwd=cs, capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for c in rows:
    if filt and filt not in c["name"]:
        continue
    print(f'{c["name"][:90]:90s} vgpr {c.get("VGPRs", -1):3d} agpr {c.get("AGPRs", 0):3d} spill {c.get("VGPRs Spill", 0):3d} scratch {c.get("ScratchSize", 0):4d} '
          f'sgpr {c.get("TotalSGPRs", 0):3d} occ {c.get("Occupancy", 0)} lds {c.get("LDS Size", 0)}')
if r.returncode:
    print(r.stderr[-3000:])
