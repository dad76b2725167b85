#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, the TCC block cannot
hold both) of the bench command.

    tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.txt> <out.json>

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports KiB and
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE * 1024 as it is.  The json maps the kernel names bench.py uses to bytes per launch (mean over dispatches).
When the passes ran `bench.py --train-only` (a process that executes training steps and nothing else; the tuner's trial
launches are kept out by LH_TUNE_CACHE), the json also carries "__train_step__": the bytes ALL kernels moved, divided by
the number of steps the process executed (= launches of the once-per-step Adam kernel) -- bench.py's
roofline.step_traffic."""
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
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
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
    if train_only:
        rd_all = sum(rd * n for _, n, rd, _, _ in rows)
        wr_all = sum(wr * write.get(k, (0.0, 0))[1] for k, _, _, wr, _ in rows)
        out["__train_step__"] = {"bytes": round((rd_all + wr_all) / steps), "read_bytes": round(rd_all / steps),
                                 "write_bytes": round(wr_all / steps), "steps": steps,
                                 "note": "all kernels of the process / launches of the once-per-step Adam kernel; valid for a "
                                         "`bench.py --train-only` process only (an inference graph in the process would be counted in)"}
        with open(sys.argv[3], "a") as f:
            f.write(f"\nWhole process: {rd_all / 1e9:.2f} GB read + {wr_all / 1e9:.2f} GB written over {steps} training steps = "
                    f"{(rd_all + wr_all) / steps / 1e9:.2f} GB per step (bench.py --train-only).\n")
    json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
