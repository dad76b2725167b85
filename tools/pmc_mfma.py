"""MFMA utilisation of the convolution kernels from a rocprofv3 PMC run of tools/conv_bench.py.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d DIR -o m -- \
        python3 tools/conv_bench.py 256 256 3 1 64 64 64 bf16 3
    python tools/pmc_mfma.py DIR/.../m_counter_collection.csv out.txt [GFLOP per launch]

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles); kernel cycles = duration x the shader clock
the run held (MI355X guide: price against the clock the chip holds under the load), taken as the highest
SQ_BUSY_CU_CYCLES / 256 CUs / duration among the kernels (a kernel whose CUs idle in its tail under-counts).
"""
import csv
import re
import sys
from collections import defaultdict


def main(path, out, gflop=309.24):
    rows = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "ring_kernel" not in k:
            continue
        m = re.match(r"_Z\d+(\w+?_kernel)I(DF16b|DF16_|Dh|f)((?:Li\d+E)+)E", k)      # mangled when the trace has no demangler
        if m:
            k = f"{m.group(1)}<{ {'DF16b': 'bf16', 'DF16_': 'f16', 'Dh': 'f16', 'f': 'float'}[m.group(2)]}," + ",".join(re.findall(r"Li(\d+)E", m.group(3))) + ">"
        k = k.replace("void ", "").split("(")[0].replace("__bf16", "bf16").replace(" ", "")
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[k]["_dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = ["rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python3 tools/conv_bench.py 256 256 3 1 64 64 64 bf16 3",
             f"(3x3 convolution 256 -> 256 channels on 64 x 64 x 64 pixels, bf16: {gflop} GFLOP per launch; mean per dispatch, durations",
             " as seen under the counter run: dispatches are serialised and run slower than in the un-profiled bench)", "",
             f"{'kernel':66s} {'launches':>8s} {'dur us':>8s} {'MFMA busy':>11s} {'CU busy':>11s} {'clk GHz':>8s} {'MFMA util':>9s} {'TFLOP/s':>8s}"]
    mean = lambda v: sum(v) / len(v)
    rows = {k: c for k, c in rows.items() if c["SQ_VALU_MFMA_BUSY_CYCLES"] and mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) > 0}
    clk = max(mean(c["SQ_BUSY_CU_CYCLES"]) / 256 / (mean(c["_dur"]) * 1e-6) / 1e9 for c in rows.values())
    for k, c in sorted(rows.items()):
        n = len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        dur, mf, cu = mean(c["_dur"]), mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(c["SQ_BUSY_CU_CYCLES"])
        util = mf / (1024 * clk * 1e9 * dur * 1e-6)
        lines.append(f"{k:66s} {n:8d} {dur:8.1f} {mf:11.3e} {cu:11.3e} {clk:8.2f} {100 * util:8.1f}% {gflop / dur * 1e3:8.0f}")
    lines += ["", "MFMA busy = 16 cycles per v_mfma_f32_16x16x32_bf16 per SIMD; utilisation = busy / (1024 SIMDs x kernel cycles at the",
              "measured clock).  TFLOP/s here is under the profiler; un-profiled rates are in gpurun_out/sweep*.txt and DESIGN.md 3.1."]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 309.24)
