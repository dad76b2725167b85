"""Times the training stem's pool passes on the benchmark shape (64 x 128 x 128 x 64, bf16): lh_bn_relu_maxpool3x3s2_fwd and
lh_maxpool3x3s2_bwd_gated.  usage (GPU box): [LH_POOL_STRIP=0|2|4] python tools/pool_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lighthand_amd import _lib

lib = _lib.load()
n, h, w, c = 64, 128, 128, 64
ho, wo = h // 2, w // 2
torch.manual_seed(0)
raw = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)
scale = (0.5 + torch.rand(c, device="cuda"))
shift = 0.3 * torch.randn(c, device="cuda")
out = torch.empty(n, ho, wo, c, dtype=torch.bfloat16, device="cuda")
idx = torch.empty(n, ho, wo, c, dtype=torch.uint8, device="cuda")
flush = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        flush.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


fwd = lambda: _lib.check(lib.lh_bn_relu_maxpool3x3s2_fwd(raw.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), idx.data_ptr(), n, h, w, c, _lib.LH_BF16, s))
fwd()
print(f"LH_POOL_STRIP={os.environ.get('LH_POOL_STRIP', '-')}: bn + relu + maxpool fwd {timed(fwd):.1f} us (cold input), checksum {int(idx.sum())} {float(out.float().sum()):.3f}")

dy = torch.randn(n, ho, wo, c, device="cuda").to(torch.bfloat16)
dx = torch.empty(n, h, w, c, dtype=torch.bfloat16, device="cuda")
mean, invstd = 0.1 * torch.randn(c, device="cuda"), 0.5 + torch.rand(c, device="cuda")
rows = lib.lh_maxpool3x3s2_bwd_gated_rows(n, h, w, c, _lib.LH_BF16)
partial = torch.zeros(rows, 2, c, device="cuda")
gate = _lib.BnBwdGate(raw.data_ptr(), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), partial.data_ptr())
bwd = lambda: _lib.check(lib.lh_maxpool3x3s2_bwd_gated(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), C.byref(gate), n, h, w, c, _lib.LH_BF16, s))
bwd()
print(f"LH_POOL_BLOCK={os.environ.get('LH_POOL_BLOCK', '-')}: maxpool bwd + BN-backward gate {timed(bwd):.1f} us (cold input), checksum {float(dx.float().sum()):.3f} {float(partial.sum()):.3f}")
