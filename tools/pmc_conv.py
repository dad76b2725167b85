#!/usr/bin/env python3
"""Hardware counters of ONE convolution launch under chosen kernel configurations: what holds the matrix pipe.

    run (under rocprofv3, one --pmc pass per counter group; the box refuses --pmc together with the tracing domains other
    than --kernel-trace):
        rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d DIR -o p -- \
            python3 tools/pmc_conv.py run CIN COUT K STRIDE N H W TRANSPOSED prec "bm,bp,depth,kb;bm,bp,depth,kb;..."
    report (any number of counter CSVs of such passes):
        python tools/pmc_conv.py report out.txt GFLOP DIR1/..._counter_collection.csv DIR2/... ...

`run` builds the layer through the engine, forces each listed configuration in turn on the forward launch and runs it
a few times (cold caches are not emulated: back-to-back launches).  `report` prints, per kernel instantiation, the mean of
every counter per dispatch and the ratios the MI355X guide defines: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x
SQ_BUSY_CU_CYCLES); LDS array busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES, bank-conflict share = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE; wave time parked at waits / barriers = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue stalls = SQ_WAIT_INST_ANY
/ SQ_WAVE_CYCLES (LDS part: SQ_WAIT_INST_LDS); L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS)."""
import csv
import os
import re
import sys
from collections import defaultdict


def run(argv):
    import ctypes as C
    import torch
    import torch.nn as nn
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["LH_AUTOTUNE"] = "0"
    from lighthand_amd import _lib
    from lighthand_amd.module import HipModule
    cin, cout, k, s, n, h, w, tr = map(int, argv[:8])
    prec = argv[8]
    cfgs = [tuple(int(v) for v in c.split(",")) for c in argv[9].split(";")]
    iters = int(argv[10]) if len(argv) > 10 else 5

    class Net(HipModule):
        def __init__(self):
            super().__init__()
            self.conv = nn.ConvTranspose2d(cin, cout, k, 2, 1, 0, bias=False) if tr else nn.Conv2d(cin, cout, k, s, k // 2, bias=False)

        def describe(self, gb):
            x = gb.input_act(cin)
            gb.output(gb.deconv(x, "conv", k) if tr else gb.conv(x, "conv", k, s, k // 2))

    lib = _lib.load()
    m = Net().cuda().set_precision(prec)
    plan = m.plan(n, h, w, training=False, backward=False)
    plan.in_act.buf.normal_()
    sp = torch.cuda.current_stream().cuda_stream
    plan.refresh_packs(sp)
    call = [c for c in plan.fwd if getattr(c, "fn", None) in (lib.lh_igemm, lib.lh_igemm_phases)][0]
    ds = call.keep if isinstance(call.keep, list) else [call.keep]
    for cfg in cfgs:
        for d in ds:
            for i in range(4):
                d.cfg[i] = cfg[i]
        for _ in range(iters):
            call(sp)
        torch.cuda.synchronize()
    print("ran", cfgs)


def demangle(k):
    m = re.match(r"_Z\d+(\w+?_kernel)I(DF16b|DF16_|Dh|f)((?:Li\d+E)+)E", k)
    if m:
        return f"{m.group(1)}<{ {'DF16b': 'bf16', 'DF16_': 'f16', 'Dh': 'f16', 'f': 'float'}[m.group(2)]}," + ",".join(re.findall(r"Li(\d+)E", m.group(3))) + ">"
    return k.replace("void ", "").split("(")[0].replace("__bf16", "bf16").replace(" ", "")


def report(argv):
    out, gflop, paths = argv[0], float(argv[1]), argv[2:]
    rows = defaultdict(lambda: defaultdict(list))
    for p in paths:
        seen = set()
        for r in csv.DictReader(open(p)):
            k = demangle(r["Kernel_Name"])
            if "igemm" not in k and "conv3x3" not in k and "wgrad_ring" not in k:
                continue
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                rows[k]["_dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    mean = lambda v: sum(v) / max(len(v), 1)
    lines = [f"counters per dispatch (mean), {gflop} GFLOP per launch; durations as seen under the counter passes", ""]
    for k, c in sorted(rows.items(), key=lambda kv: mean(kv[1]["_dur"])):
        g = lambda n: mean(c[n]) if c.get(n) else None
        dur = mean(c["_dur"])
        lines.append(f"{k}: {dur:.1f} us under the profiler = {gflop / dur * 1e3 / 1e3:.0f} TFLOP/s")
        cu, wc = g("SQ_BUSY_CU_CYCLES"), g("SQ_WAVE_CYCLES")
        der = []
        if cu:
            der.append(f"clock held {cu / 256 / (dur * 1e-6) / 1e9:.2f} GHz")
            if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                der.append(f"MFMA busy {100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * cu):.1f} %")
            if g("SQ_LDS_IDX_ACTIVE") is not None:
                der.append(f"LDS array busy {100 * g('SQ_LDS_IDX_ACTIVE') / cu:.1f} % of CU-busy cycles")
        if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
            der.append(f"bank-conflict cycles {100 * g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.1f} % of LDS-active")
        if wc:
            for nm, lab in (("SQ_WAIT_ANY", "parked at s_waitcnt / barrier"), ("SQ_WAIT_INST_ANY", "issue stalls"),
                            ("SQ_WAIT_INST_LDS", "issue stalls on LDS"), ("SQ_ACTIVE_INST_ANY", "issuing"),
                            ("SQ_ACTIVE_INST_LDS", "issuing LDS"), ("SQ_ACTIVE_INST_VMEM", "issuing VMEM"),
                            ("SQ_ACTIVE_INST_VALU", "issuing VALU / MFMA"), ("SQ_ACTIVE_INST_SCA", "issuing scalar"),
                            ("SQ_INST_CYCLES_VMEM", "VMEM instruction cycles")):
                if g(nm) is not None:
                    der.append(f"{lab} {100 * g(nm) / wc:.1f} % of wave cycles")
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
            der.append(f"L2 hit rate {100 * g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1):.1f} %")
        lines.append("    " + "; ".join(der))
        lines.append("    " + "  ".join(f"{n}={mean(v):.4g}" for n, v in sorted(c.items()) if not n.startswith("_")))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    (run if sys.argv[1] == "run" else report)(sys.argv[2:])
