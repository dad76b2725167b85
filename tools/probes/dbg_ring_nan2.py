#!/usr/bin/env python3
"""Debug: tests/test_gpu_ops.py::test_direct3x3_kernel_bn_statistics[case1] on poisoned allocator memory, run the way the test runs
it (plan.run_forward / run_backward with their side streams); reports NaNs per result tensor."""
import copy, os, sys
import torch, torch.nn as nn
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from lighthand_amd.engine import Plan
from lighthand_amd.module import HipModule
import test_gpu_ops as T

mode = sys.argv[1] if len(sys.argv) > 1 else "poison"
if mode == "poison":
    junk = []
    for rep in range(3):
        for e in range(9, 27):
            for mul in (1.0, 1.5):
                junk.append(torch.full((int((1 << e) * mul) // 4,), float("nan"), device="cuda"))
    torch.cuda.synchronize(); del junk
for case in ((64, 64, 2, 16, 16), (32, 32, 3, 12, 20)):
    cin, cout, n, h, w = case
    class Net(HipModule):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
            self.bn = nn.BatchNorm2d(cout, momentum=0.1)
            self.out = nn.Conv2d(cout, 8, 1, bias=False)
        def describe(self, gb):
            x = gb.input_act(cin)
            gb.output(gb.conv(gb.fuse([(gb.conv(x, "conv", 3, 1, 1), "bn")]), "out", 1, 1, 0))
    torch.manual_seed(23)
    proto = Net()
    x = torch.randn(n, cin, h, w).to(torch.bfloat16).float()
    for which in ("tiled", "direct"):
        Plan.force_cfg = (lambda cands: next(c for c in cands if c[2] not in (1, 100))) if which == "tiled" else \
                         (lambda cands: next((c for c in cands if c[2] == 100), cands[0]))
        m = copy.deepcopy(proto)
        torch.manual_seed(24)
        out, dx, grads = T._run_plan(m, x, lambda o: torch.randn_like(o), "bf16")
        plan = next(iter(m._lh_plans.values()))
        print(case, which, "out nan", int(torch.isnan(out).sum()), "dx nan", int(torch.isnan(dx).sum()), "of", dx.numel(),
              {k: int(torch.isnan(v).sum()) for k, v in grads.items()},
              [(getattr(c, "what", "")[:28], tuple(c.keep.cfg[:4])) for c in plan.bwd if hasattr(getattr(c, "keep", None), "cfg")])
