#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, the TCC block cannot
hold both) of the bench command.

    tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.txt> <out.json>

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports KiB and
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE * 1024 as it is.  The json maps the kernel names bench.py uses to bytes per launch (mean over dispatches).
When the passes ran `bench.py --train-only` (a process that executes training steps and nothing else; the tuner's trial
launches are kept out by LH_TUNE_CACHE), the json also carries "__train_step__": the bytes ALL kernels moved BEHIND the first launch of the
once-per-step Adam kernel (everything in front of it is plan construction -- zero fills of the buffers, pack setup -- and the
first, eager step), divided by the steps that follow it (= Adam launches - 1) -- bench.py's roofline.step_traffic.  (Round 5
divided the whole process by all steps: ~2 % of construction fills counted as step traffic.)"""
import collections
import csv
import json
import re
import sys

TYPES = {"DF16b": "__bf16", "DF16_": "_Float16", "f": "float"}


def demangle(name):
    m = re.match(r"_Z\d+([A-Za-z_0-9]+?)I(DF16b|DF16_|f)((?:Li\d+E)*)E", name)
    if not m:
        return name
    ints = re.findall(r"Li(\d+)E", m.group(3))
    return f"{m.group(1)}<{', '.join([TYPES[m.group(2)]] + ints)}>"


def load(path, counter):
    """{kernel: [sum, dispatches]} and the steady-state share: (sum over the dispatches behind the first Adam launch, Adam launches behind it)."""
    acc = collections.defaultdict(lambda: [0.0, 0])
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    adam = [d for d, k, _ in rows if "adam_kernel" in k]
    steady = (0.0, 0)
    if len(adam) >= 2:
        steady = (sum(v for d, _, v in rows if adam[0] < d <= adam[-1]), len(adam) - 1)
    return acc, steady


def main():
    (fetch, fsteady), (write, wsteady) = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k, (v, n) in fetch.items():
        w, wn = write.get(k, (0.0, 0))
        rows.append((k, n, 2 * v * 1024 / n, (w * 1024 / wn) if wn else 0.0, 2 * v * 1024 + (w * 1024 if wn else 0)))
    rows.sort(key=lambda r: -r[4])
    with open(sys.argv[3], "w") as f:
        f.write("PMC passes of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only` (rocprofv3 --kernel-trace "
                "--pmc FETCH_SIZE / WRITE_SIZE, separate runs).\nUnits: rocprofv3 reports KiB; per the MI355X guide FETCH_SIZE "
                "under-counts wide coalesced reads by 2x on gfx950, so read bytes = 2 * FETCH_SIZE * 1024.\nPer-dispatch means, "
                "sorted by total bytes moved.\n\n")
        f.write(f"{'kernel':92s} {'launches':>8s} {'read MB/launch (x2 corrected)':>30s} {'write MB/launch':>16s}\n")
        for k, n, rd, wr, _ in rows[:40]:
            f.write(f"{demangle(k)[:92]:92s} {n:8d} {rd / 1e6:30.2f} {wr / 1e6:16.2f}\n")
    out = {demangle(k): {"read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "launches": n}
           for k, n, rd, wr, _ in rows}
    steps = sum(n for k, n, *_ in rows if "adam_kernel" in k)
    # a training step decodes its heat-maps once; a process that also replays the inference graph launches the arg-max more often
    train_only = steps > 0 and sum(n for k, n, *_ in rows if "heatmap_argmax" in k) == steps
    if train_only and fsteady[1] >= 1 and fsteady[1] == wsteady[1]:
        n_st = fsteady[1]
        rd_all, wr_all = 2 * fsteady[0] * 1024, wsteady[0] * 1024
        whole = (sum(rd * n for _, n, rd, _, _ in rows) + sum(wr * write.get(k, (0.0, 0))[1] for k, _, _, wr, _ in rows)) / steps
        out["__train_step__"] = {"bytes": round((rd_all + wr_all) / n_st), "read_bytes": round(rd_all / n_st),
                                 "write_bytes": round(wr_all / n_st), "steps": n_st,
                                 "whole_process_bytes_per_step": round(whole),
                                 "note": "all kernels dispatched behind the first Adam launch / the Adam launches behind it (plan construction and the "
                                         "first eager step excluded); valid for a `bench.py --train-only` process only (an inference graph in the "
                                         "process would be counted in).  whole_process_bytes_per_step = round 5's figure (everything / all steps)"}
        with open(sys.argv[3], "a") as f:
            f.write(f"\nSteady state (behind the first Adam launch): {rd_all / 1e9:.2f} GB read + {wr_all / 1e9:.2f} GB written over {n_st} training steps = "
                    f"{(rd_all + wr_all) / n_st / 1e9:.2f} GB per step (bench.py --train-only); the whole process over all {steps} steps: {whole / 1e9:.2f} GB per step.\n")
    json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
