"""Merged 4-branch 3x3 launch (HRNet-W32 stage 4, bs 32): real shapes vs stand-ins with the deep-K members' K split
(K / s with s x the pixels: what a split-K form of those members would run, minus its reduction)."""
import ctypes as C, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lighthand_amd import _lib
from lighthand_amd._lib import check
from lighthand_amd.engine import _desc
lib = _lib.load()
dt = _lib.LH_BF16
taps = [(r - 1, q - 1) for r in range(3) for q in range(3)]
def prob(n, h, w, cin, cout):
    return _desc(n, h, w, cin, cin, h, w, 1, 1, cout, h, w, 1, 1, 0, 0, cout, taps)
def bufs(d):
    es = 2
    kpad = (d.k_run * es + 127) // 128 * 128
    src = torch.randn(d.n * d.hi * d.wi * d.in_pix_stride, device="cuda").to(torch.bfloat16)
    pack = (torch.randn((d.cout + 255) // 256 * 256 * d.ntaps * kpad // 2, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(d.n * d.OH * d.OW * d.out_pix_stride, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(max((d.n * d.ho * d.wo + 63) // 64, 1024) * 2 * d.cout, device="cuda")
    return src, pack, out, stats
def timed(ds, cfg, stats=True, iters=30):
    arr = (_lib.IgemmCall * len(ds))()
    keep = []
    for i, d in enumerate(ds):
        d.cfg[0], d.cfg[1], d.cfg[2], d.cfg[3] = cfg
        s, p, o, st = bufs(d)
        keep.append((s, p, o, st))
        arr[i].d = C.pointer(d); arr[i].in_ = s.data_ptr(); arr[i].wpack = p.data_ptr(); arr[i].out = o.data_ptr()
        if stats: arr[i].stats = st.data_ptr()
    sp = torch.cuda.current_stream().cuda_stream
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    ts = []
    for it in range(iters + 3):
        flush.fill_(it & 1)
        for s, _, _, _ in keep: s.add_(0)          # operands back into the caches, like a producer kernel leaves them
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.lh_igemm_multi(arr, len(ds), dt, sp), "multi")
        e1.record(); torch.cuda.synchronize()
        if it >= 3: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
B = 32
real = lambda: [prob(B, 64, 64, 32, 32), prob(B, 32, 32, 64, 64), prob(B, 16, 16, 128, 128), prob(B, 8, 8, 256, 256)]
emu2 = lambda: [prob(B, 64, 64, 32, 32), prob(B, 32, 32, 64, 64), prob(2 * B, 16, 16, 64, 128), prob(4 * B, 8, 8, 64, 256)]
emu3 = lambda: [prob(B, 64, 64, 32, 32), prob(B, 32, 32, 64, 64), prob(4 * B, 16, 16, 32, 128), prob(8 * B, 8, 8, 32, 256)]
for cfg in [(64, 128, 2, 64), (64, 128, 4, 64), (64, 64, 2, 64), (64, 64, 4, 64)]:
    print(cfg, "real 4x %.1f us | b2 K/2, b3 K/4: %.1f us | b2 K/4, b3 K/8: %.1f us | b0 alone %.1f | b3 alone %.1f | b0+b1 %.1f" % (
        timed(real(), cfg), timed(emu2(), cfg), timed(emu3(), cfg), timed(real()[:1], cfg), timed(real()[3:], cfg), timed(real()[:2], cfg)), flush=True)
