"""Debug: per-parameter gradient difference between the fused (lh_igemm_bntap) and separate BN-backward reduce."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_gpu_model import _build          # noqa: E402
from lighthand_amd.heatmap import JointsMSELoss   # noqa: E402


def grads(tag, prec, notap, noflat=False):
    if noflat:
        os.environ["LH_NO_FLAT"] = "1"
    else:
        os.environ.pop("LH_NO_FLAT", None)
    if notap:
        os.environ["LH_NO_BNTAP"] = "1"
    else:
        os.environ.pop("LH_NO_BNTAP", None)
    torch.manual_seed(0)
    model, _ = _build(tag)
    model = model.cuda().train().set_precision(prec)
    rng = np.random.RandomState(3)
    x = torch.from_numpy(rng.randn(4, 3, 64, 64).astype(np.float32)).cuda()
    out = model(x)
    JointsMSELoss(False)(out, torch.zeros_like(out), None).backward()
    return {k: p.grad.detach().double().cpu().numpy().copy() for k, p in model.named_parameters()}


for tag, prec in [("mini_bottleneck", "bf16"), ("r50", "bf16")]:
    a, b, c = grads(tag, prec, False), grads(tag, prec, True), grads(tag, prec, True, noflat=True)
    rows = []
    for k in b:
        e = np.abs(a[k] - b[k]).max() / (np.abs(b[k]).max() + 1e-20)
        l2 = np.linalg.norm(a[k] - b[k]) / (np.linalg.norm(b[k]) + 1e-20)
        det = np.abs(c[k] - b[k]).max() / (np.abs(b[k]).max() + 1e-20)
        rows.append((e, l2, det, k))
    rows.sort(reverse=True)
    print(tag, prec, "worst:", [(f"{e:.2e}", f"{l2:.2e}", f"{d:.1e}", k) for e, l2, d, k in rows[:4]], "median", f"{np.median([r[0] for r in rows]):.2e}")
