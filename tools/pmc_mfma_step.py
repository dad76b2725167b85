#!/usr/bin/env python3
"""MFMA utilisation of EVERY kernel of the bench step, at the shapes the step runs, from one rocprofv3 PMC pass:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d DIR -o m -- \
        python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra
    python tools/pmc_mfma_step.py DIR/.../m_counter_collection.csv out.txt

Per kernel name (mean over its dispatches): duration under the counter run, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /
(4 SIMDs x SQ_BUSY_CU_CYCLES) -- the share of the matrix pipes' cycles at the clock the kernel held -- the same busy
cycles against 1024 SIMDs x 2.4 GHz x duration, and the clock the kernel held (CU-busy cycles / 256 CUs / duration; a
kernel that leaves CUs idle under-counts it)."""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import demangle


def main(path, out, title):
    rows = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = demangle(r["Kernel_Name"]).replace("void ", "").split("(")[0]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
            rows[k]["_dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    mean = lambda v: sum(v) / max(len(v), 1)
    tab = []
    for k, c in rows.items():
        n, dur = len(c["_dur"]), mean(c["_dur"])
        mf, cu = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(c["SQ_BUSY_CU_CYCLES"])
        if n == 0 or dur <= 0:
            continue
        tab.append((n * dur, k, n, dur, 100 * mf / (4 * cu) if cu else 0.0, 100 * mf / (1024 * 2.4e9 * dur * 1e-6), cu / 256 / (dur * 1e-6) / 1e9))
    tab.sort(reverse=True)
    total = sum(t[0] for t in tab)
    lines = [title, "(mean per dispatch; dispatches are serialised under the counter run and run slower than in the un-profiled bench)", "",
             f"{'kernel':72s} {'launches':>8s} {'dur us':>8s} {'share':>6s} {'MFMA busy':>10s} {'@2.4 GHz':>9s} {'clk GHz':>8s}"]
    for tt, k, n, dur, util, util24, clk in tab[:60]:
        lines.append(f"{k[:72]:72s} {n:8d} {dur:8.1f} {100 * tt / total:5.1f}% {util:9.1f}% {util24:8.1f}% {clk:8.2f}")
    mm = sum(t[0] * t[4] for t in tab if "igemm" in t[1] or "wgrad_ring" in t[1]) / max(sum(t[0] for t in tab if "igemm" in t[1] or "wgrad_ring" in t[1]), 1e-9)
    lines += ["", f"time-weighted MFMA busy of the convolution kernels (igemm_*, wgrad_ring_*): {mm:.1f} % of the matrix pipes' cycles",
              "MFMA busy counts 16 cycles per v_mfma_f32_16x16x32 per SIMD."]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra")
