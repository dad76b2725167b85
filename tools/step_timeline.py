#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`, the *_kernel_trace.csv
of tools/refresh_profiles.sh): per hardware queue the number of kernels, their summed duration and when the queue was
active; the time during which ANY kernel ran; idle gaps; the serial tail after the last main-stream kernel.
usage: step_timeline.py <kernel_trace.csv> [out.txt]   (the step analysed = the graph replay in the middle of the run,
delimited by the Adam launches)"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    ad = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
    k = len(ad) // 2
    seg = rows[ad[k] + 1:ad[k + 1] + 1]
    t0 = min(int(r["Start_Timestamp"]) for r in seg)
    t1 = max(int(r["End_Timestamp"]) for r in seg)
    us = lambda ns: ns / 1e3
    print(f"step = dispatches between Adam launch {k} and {k + 1} of {len(ad)}: {len(seg)} kernels, {us(t1 - t0):.1f} us from the first "
          f"kernel's start to the end of Adam", file=out)
    qs = {}
    for r in seg:
        qs.setdefault(r["Queue_Id"], []).append(r)
    print(f"{'queue':>5} {'kernels':>8} {'sum of durations us':>20} {'first start us':>15} {'last end us':>12}", file=out)
    for q, l in sorted(qs.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in l)
        print(f"{q:>5} {len(l):>8} {us(busy):>20.1f} {us(min(int(r['Start_Timestamp']) for r in l) - t0):>15.1f} "
              f"{us(max(int(r['End_Timestamp']) for r in l) - t0):>12.1f}", file=out)
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    cs, ce = ev[0]
    union, gaps = 0, []
    for s, e in ev[1:]:
        if s > ce:
            union += ce - cs
            gaps.append((s - ce, ce - t0))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    total = sum(e - s for s, e in ev)
    print(f"sum of all kernel durations {us(total):.1f} us; some kernel running for {us(union):.1f} us (overlap factor "
          f"{total / union:.2f}); idle {us(t1 - t0 - union):.1f} us in {len(gaps)} gaps, largest "
          f"{', '.join(f'{us(g):.1f} us at {us(a):.0f}' for g, a in sorted(gaps, reverse=True)[:5])}", file=out)
    mainq = max(qs, key=lambda q: len(qs[q]))
    order = sorted(seg, key=lambda r: int(r["Start_Timestamp"]))
    # what borders the largest gaps: the kernels that ended last before each and the ones that started first after it
    row = lambda r: (f"  q{r['Queue_Id']} {us(int(r['Start_Timestamp']) - t0):>9.1f} us  "
                     f"{us(int(r['End_Timestamp']) - int(r['Start_Timestamp'])):>7.1f} us  {r['Kernel_Name'][:90]}")
    for g, a in sorted(gaps, reverse=True)[:8]:
        print(f"gap of {us(g):.1f} us at {us(a):.0f} us: the two kernels that ended last before it, the three that started first after it", file=out)
        before = sorted((r for r in seg if int(r["End_Timestamp"]) - t0 <= a), key=lambda r: int(r["End_Timestamp"]))[-2:]
        after = [r for r in order if int(r["Start_Timestamp"]) - t0 >= a + g][:3]
        for r in before:
            print(row(r), file=out)
        print("    ...", file=out)
        for r in after:
            print(row(r), file=out)
    print(f"the first 30 kernels of the step:", file=out)
    for r in order[:30]:
        print(row(r), file=out)
    print(f"the last 14 kernels of the step (queue {mainq} = the main stream's):", file=out)
    for r in order[-14:]:
        print(f"  q{r['Queue_Id']} {us(int(r['Start_Timestamp']) - t0):>9.1f} us  {us(int(r['End_Timestamp']) - int(r['Start_Timestamp'])):>7.1f} us  "
              f"{r['Kernel_Name'][:90]}", file=out)


if __name__ == "__main__":
    main()
