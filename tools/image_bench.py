"""Times lh_image_to_nhwc4 on the benchmark batch (64 x 3 x 256 x 256 fp32 -> padded NHWC4 bf16).  usage (GPU box): python tools/image_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lighthand_amd import _lib
lib = _lib.load()
n, h, w, pad = 64, 256, 256, 3
hp, wp = h + 2 * pad, w + 2 * pad + 2
x = torch.randn(n, 3, h, w, device="cuda")
out = torch.empty(n, hp, wp, 4, dtype=torch.bfloat16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
f = lambda: _lib.check(lib.lh_image_to_nhwc4(x.data_ptr(), out.data_ptr(), n, h, w, pad, wp, _lib.LH_BF16, s))
f(); ts = []
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print(f"image_to_nhwc4: {ts[len(ts)//2]:.1f} us, checksum {float(out.float().sum()):.4f}")
