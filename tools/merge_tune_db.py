#!/usr/bin/env python3
"""Merge a freshly measured tuning file into the shipped database, key by key.
usage: merge_tune_db.py old_db new_measured out [policy]
policy 'dense' (default): where both files hold a key, the new choice is taken only if it names one of the configurations
the old database could not know (the dense-wave forms, ring depth 20..29); otherwise the old choice stays (near-ties fall
differently from run to run, and round 3's choices were checked to be no slower).  Keys only one file holds are kept."""
import ast
import sys


def load(path):
    head, d = [], {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            head.append(line)
            continue
        try:
            k, v = ast.literal_eval(line)
            d[k] = tuple(v)
        except (ValueError, SyntaxError, TypeError):
            pass
    return head, d


def is_new_kind(v):
    def dense(c):
        return isinstance(c, tuple) and len(c) >= 4 and all(isinstance(x, int) for x in c[:4]) and 20 <= c[2] < 30
    return dense(v) or any(dense(x) for x in v if isinstance(x, tuple))


old_head, old = load(sys.argv[1])
_, new = load(sys.argv[2])
out, taken, added = dict(old), 0, 0
for k, v in new.items():
    if k not in old:
        out[k] = v
        added += 1
    elif v != old[k] and is_new_kind(v):
        out[k] = v
        taken += 1
with open(sys.argv[3], "w") as f:
    f.writelines(h for h in old_head if h.startswith("#") and not h.startswith("# lib"))
    for k, v in out.items():
        f.write(repr((k, v)) + "\n")
print(f"{len(old)} old, {len(new)} new -> {len(out)} entries: {taken} replaced by a dense-wave choice, {added} added")
