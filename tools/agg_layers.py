#!/usr/bin/env python3
"""Side-by-side per-kernel totals (ms) of several tools/layer_profile.py outputs: agg_layers.py a.txt b.txt ..."""
import collections
import re
import sys

tabs = []
for f in sys.argv[1:]:
    t = collections.OrderedDict()
    for ln in open(f):
        m = re.match(r'(fwd|bwd) (.{42}) (.{40}) +([\d.]+) ', ln)
        if m:
            t[m.group(3).strip()] = t.get(m.group(3).strip(), 0.0) + float(m.group(4))
    tabs.append(t)
keys = sorted(tabs[0], key=lambda k: -tabs[0][k])
print(f"{'kernel':44s}" + "".join(f"{f.split('/')[-1][:12]:>13s}" for f in sys.argv[1:]))
for k in keys:
    print(f"{k:44s}" + "".join(f"{t.get(k, 0):13.3f}" for t in tabs))
print(f"{'TOTAL':44s}" + "".join(f"{sum(t.values()):13.3f}" for t in tabs))
