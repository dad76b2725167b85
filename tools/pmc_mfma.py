"""MFMA utilisation of the convolution kernels from a rocprofv3 PMC run of tools/conv_bench.py.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d DIR -o m -- \
        python3 tools/conv_bench.py 256 256 3 1 64 64 64 bf16 3
    python tools/pmc_mfma.py DIR/.../m_counter_collection.csv out.txt [GFLOP per launch]

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles); kernel cycles = duration x the shader clock
that kernel held = its SQ_BUSY_CU_CYCLES / 256 CUs / duration (MI355X guide: price against the clock the chip holds
under the load; MFMA-heavy kernels clock lower than light ones).  The same busy cycles against the 2.4 GHz peak clock are
printed beside it.  The shape is not in the shipped tuning database, so the plan's tuner runs under the profiler: every
candidate appears with its trial launches; the configuration the plan chose is the one with the most launches (*).
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
             f"{'kernel':50s} {'launches':>8s} {'dur us':>8s} {'MFMA busy':>11s} {'CU busy':>11s} {'clk GHz':>8s} {'MFMA util':>9s} {'@2.4 GHz':>9s} {'TFLOP/s':>8s}"]
    mean = lambda v: sum(v) / len(v)
    rows = {k: c for k, c in rows.items() if c["SQ_VALU_MFMA_BUSY_CYCLES"] and mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) > 0}
    chosen = {}
    for k, c in rows.items():
        fam = k.split("<")[0]
        if fam not in chosen or len(c["SQ_VALU_MFMA_BUSY_CYCLES"]) > len(rows[chosen[fam]]["SQ_VALU_MFMA_BUSY_CYCLES"]):
            chosen[fam] = k
    for k, c in sorted(rows.items(), key=lambda kv: (kv[0].split("<")[0], mean(kv[1]["_dur"]))):
        n = len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        dur, mf, cu = mean(c["_dur"]), mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(c["SQ_BUSY_CU_CYCLES"])
        clk = cu / 256 / (dur * 1e-6) / 1e9
        util = mf / (4 * cu)
        util24 = mf / (1024 * 2.4e9 * dur * 1e-6)
        mark = " *" if chosen[k.split("<")[0]] == k else ""
        lines.append(f"{k + mark:50s} {n:8d} {dur:8.1f} {mf:11.3e} {cu:11.3e} {clk:8.2f} {100 * util:8.1f}% {100 * util24:8.1f}% {gflop / dur * 1e3:8.0f}")
    ref = max(mean(rows[k]["SQ_BUSY_CU_CYCLES"]) / 256 / (mean(rows[k]["_dur"]) * 1e-6) / 1e9 for k in chosen.values())
    lines += ["", f"Conservative reading (a kernel that leaves CUs idle under-counts its CU-busy cycles): the chosen kernels against the highest clock among them, {ref:.2f} GHz:"]
    for k in chosen.values():
        c = rows[k]
        lines.append(f"  {k:48s} {100 * mean(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024 * ref * 1e9 * mean(c['_dur']) * 1e-6):5.1f}%")
    lines += ["", "MFMA busy = 16 cycles per v_mfma_f32_16x16x32_bf16 per SIMD; utilisation = busy / (1024 SIMDs x kernel cycles at the clock",
              "the kernel held).  TFLOP/s here is under the profiler; un-profiled rates are in gpurun_out/sweep*.txt and DESIGN.md 3.1.",
              "(*) the configuration the plan runs: forward and data gradient share the igemm kernel, the weight gradient has its own."]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 309.24)
