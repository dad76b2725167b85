#!/usr/bin/env python3
"""List the s_waitcnt vmcnt(0) the COMPILER placed between the first LDS-DMA (or first MFMA) and the last MFMA of every kernel
of a translation unit: each is a point where a wave waits for EVERYTHING it has in flight (ring stages, output stores).
The kernels' own waits are inline asm (between ;;#ASMSTART / ;;#ASMEND) and are not listed.
usage: tools/isa_waits.py unit.hip [unit.hip ...]      (run from lighthand_amd/csrc)"""
import re
import subprocess
import sys

for unit in sys.argv[1:]:
    out = f"/tmp/{unit.replace('/', '_')}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                    unit, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    kern, idx = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
            idx[kern] = dict(mfma=[], w=[], dma=[])
        if kern:
            if "v_mfma" in l:
                idx[kern]["mfma"].append(i)
            if "global_load_lds" in l:
                idx[kern]["dma"].append(i)
            if re.search(r"s_waitcnt.*vmcnt\(0\)", l) and "ASMSTART" not in lines[i - 1]:
                idx[kern]["w"].append(i)
    total = 0
    for k, v in idx.items():
        if not v["mfma"]:
            continue
        a, b = (v["dma"] or v["mfma"])[0], v["mfma"][-1]
        inside = [w for w in v["w"] if a < w < b]
        total += len(inside)
        if inside:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print(f"{unit}: {name[:90]}: {len(inside)} at lines {inside[:8]} of {out}")
    print(f"{unit}: {total} compiler vmcnt(0) inside the MFMA region of {sum(1 for v in idx.values() if v['mfma'])} kernels")
