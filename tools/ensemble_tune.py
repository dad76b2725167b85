#!/usr/bin/env python3
"""Ensemble of several tuning sessions: LH_TUNE_TIMES files (one line per timed candidate: key, choice, ms) of N sessions -- normally N boxes
of the pool, which favour different candidates among near-ties -- are added per (key, choice); the database entry of a key becomes the choice
with the smallest SUM over the sessions that timed it in all of them.
usage: ensemble_tune.py out_db times1.txt times2.txt ...   (keys of _tune / _tune_wgrad / _tune_table; group keys of HRNet are not logged)"""
import ast
import collections
import sys

out, files = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fi, path in enumerate(files):
    seen = set()
    for line in open(path):
        try:
            key, choice, ms = ast.literal_eval(line)
        except (ValueError, SyntaxError):
            continue
        if (key, choice) in seen:                 # a key tuned twice in one session (two plans): first timing counts
            continue
        seen.add((key, choice))
        a = acc[key][tuple(choice)]
        a[0] += ms
        a[1] += 1
n = len(files)
with open(out, "w") as f:
    kept = 0
    for key, cands in acc.items():
        full = {c: v[0] for c, v in cands.items() if v[1] == n}
        if not full:
            continue
        best = min(full, key=full.get)
        f.write(repr((key, best)) + "\n")
        kept += 1
print(f"{kept} keys from {n} sessions")
