#!/usr/bin/env python3
"""Headline benchmark: images/sec of the SimpleBaseline-ResNet50 training step (256x256, bs=64 per
GPU, bf16 activations/weights with fp32 accumulation) on the HIP engine -- BASELINE.json configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one full training iteration on a synthetic batch already resident in HBM: weight packs,
forward, Gaussian target render from joints, MSE loss + gradient, arg-max decode, backward,
[gradient all-reduce], fused Adam -- one hipGraph replay.  Rank 0 prints ONE JSON line with
throughput, the roofline of the dominant kernel (per-launch HIP-event timing, measured live) and
a CPU baseline (the oracle's plain-PyTorch restatement on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
# SURVEY.md section 8(d) / BASELINE.md section 3: algorithmic work per image -- conv/deconv FLOPs (x2 per MAC) and
# ideal-fusion activation traffic at 2 bytes per element: (fwd GFLOP, train GFLOP, fwd MB, train MB)
WORK = {
    ("r18", 256): (7.734, 22.895, 17.9, 53.8),
    ("r50", 256): (14.479, 43.128, 62.8, 188.5),
    ("r50", 384): (32.577, 97.038, 141.4, 424.1),
    ("hrnet32", 256): (20.388, 61.107, 107.2, 321.5),
    ("hrnet48", 256): (41.846, 125.483, 142.9, 428.7),
}
PMC_FILE = "r06_pmc_traffic.json"      # per-kernel HBM bytes from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py)
STATS_FILE = "r06_bench_kernel_stats.csv"   # rocprofv3 --kernel-trace --stats of this same command (profiles/README.md)
TRAIN_GFLOP_PER_IMG = 43.128
FWD_GFLOP_PER_IMG = 14.479
PEAK_BF16_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA
PEAK_F32_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA (exact fp32) = the fp32 vector rate
PEAK_HBM_GBS = 8000.0


def step_roofline(key, train, images_per_s, es=2):
    """Whole-step roofline fractions: algorithmic FLOPs vs the dense MFMA peak, algorithmic bytes vs the
    HBM peak, and the measured time against the governing (larger) of the two lower bounds."""
    w = WORK.get(key)
    if w is None or images_per_s <= 0:
        return None
    gflop, mb = (w[1], w[3]) if train else (w[0], w[2])
    mb = mb * es / 2
    t = 1.0 / images_per_s                                  # seconds per image
    peak_tf = PEAK_F32_TFLOPS if es == 4 else PEAK_BF16_TFLOPS
    t_mfma, t_hbm = gflop * 1e9 / (peak_tf * 1e12), mb * 1e6 / (PEAK_HBM_GBS * 1e9)
    return {"mfma_peak_tflops": peak_tf, "mfma_frac": round(t_mfma / t, 4), "hbm_frac": round(t_hbm / t, 4),
            "governing": "hbm" if t_hbm >= t_mfma else "mfma", "frac": round(max(t_mfma, t_hbm) / t, 4),
            "achieved_tflops": round(gflop * images_per_s / 1e3, 1), "achieved_gbs": round(mb * images_per_s / 1e3, 1),
            "gflop_per_image": gflop, "mb_per_image": mb}


def timed_replays(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def build_model(depth=50, precision="bf16", hrnet_width=0):
    import types
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    if hrnet_width:
        from lighthand_amd.modeling.hrnet.pose_hrnet import get_hrnet, hrnet_cfg
        torch.manual_seed(9001)
        return get_hrnet(hrnet_cfg(hrnet_width), True).cuda().set_precision(precision)
    ns = types.SimpleNamespace
    extra = ns(NUM_LAYERS=depth, DECONV_WITH_BIAS=False, NUM_DECONV_LAYERS=3, NUM_DECONV_FILTERS=[256] * 3,
               NUM_DECONV_KERNELS=[4] * 3, FINAL_CONV_KERNEL=1)
    torch.manual_seed(9001)                                  # src/tools/train.py:15
    return get_pose_net(ns(MODEL=ns(EXTRA=extra, STYLE="pytorch")), True).cuda().set_precision(precision)


def synthetic_batch(batch, size, device, seed=9001):
    rng = np.random.RandomState(seed)
    images = torch.from_numpy(rng.randn(batch, 3, size, size).astype(np.float32)).to(device)
    joints = torch.from_numpy(rng.uniform(20, size - 20, size=(batch, 21, 2)).astype(np.float32)).to(device)
    return images, joints


def profile_kernels(step, iters=3, plan=None, fwd_only=False):
    """Per-launch HIP-event timing of every conv-family launch of the (eager) step, on the stream the kernels are
    launched on.  An event pair around ONE launch also times event packets and the launch gap.  That overhead is estimated
    in the same pass from EMPTY pairs interleaved with the real ones: an empty pair runs two event packets back to back,
    a pair around a launch overlaps one of them with the launch, so HALF the empty-pair median is subtracted -- the value
    that makes the per-kernel averages agree with the durations `rocprofv3 --kernel-trace --stats` reports for the same
    command (calibrated on the dominant kernel: 38.9 us with the overhead vs 33.0 us in profiles/r02, 5.9 us = 0.5 x the
    empty pair).
    Returns {kernel name: dict(ms, launches, flops, bytes)} per step."""
    plan = plan or step.plan
    meta = {}
    for which, call, name, flops, nbytes in plan.profile_meta:
        meta[id(call)] = (name, flops, nbytes)
    agg = {}
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    for it in range(iters + 1):
        evs, empty = [], []
        plan.refresh_packs(s)
        for which, lst in (("fwd", plan.fwd),) if fwd_only else (("fwd", plan.fwd), ("bwd", plan.bwd)):
            if which == "bwd":
                step._fwd_loss_tail(s)
            for i, call in enumerate(lst):
                m = meta.get(id(call))
                if m is None:
                    call(s)
                    continue
                parts = _split_wgrad(plan.lib, call) if getattr(call, "fn", None) is plan.lib.lh_wgrad_fused else \
                    _split_table(plan.lib, call) if getattr(call, "fn", None) is plan.lib.lh_wgrad_table_run else None
                if parts:                                        # weight gradient and its split-K fold are two kernels: time them apart
                    for fn_, mm in zip(parts, (m, ("wgrad_reduce(kernels)", 0.0, 0.0))):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(stream)
                        fn_(s)
                        b.record(stream)
                        evs.append((mm, a, b))
                    continue
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                call(s)
                b.record(stream)
                evs.append((m, a, b))
                if len(evs) % 8 == 0:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    b.record(stream)
                    empty.append((a, b))
        torch.cuda.synchronize()
        if it == 0:
            continue                                             # first pass = warm-up
        gap = 0.5 * float(np.median([a.elapsed_time(b) for a, b in empty])) if empty else 0.0
        for (name, flops, nbytes), a, b in evs:
            d = agg.setdefault(name, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, gap_ms=0.0))
            d["ms"] += max(a.elapsed_time(b) - gap, 0.0)
            d["gap_ms"] += gap
            d["launches"] += 1
            d["flops"] += flops
            d["bytes"] += nbytes
    for d in agg.values():
        for k in ("ms", "flops", "bytes", "gap_ms"):
            d[k] /= iters
        d["launches"] //= iters
    return agg


def _split_wgrad(lib, call):
    """lh_wgrad_fused = the weight-gradient kernel, then the fold of its pixel-split slabs: the same two launches through
    their own entry points (lh_wgrad / lh_wgrad_rowfold, lh_wgrad_reduce), so that each can be timed by itself."""
    from lighthand_amd._lib import check
    d, rows, x, dy, dys, n_out, n_in, ws, grad, so, si, sr, ss, taps, acc, dt = call.args
    if rows > 1:
        first = lambda st: check(lib.lh_wgrad_rowfold(d, rows, x, dy, dys, n_out, ws, dt, st), "lh_wgrad_rowfold")
    else:
        first = lambda st: check(lib.lh_wgrad(d, x, dy, dys, n_out, n_in, ws, dt, st), "lh_wgrad")
    return first, (lambda st: check(lib.lh_wgrad_reduce(d, ws, grad, n_out, n_in, so, si, sr, ss, taps, acc, dt, st), "lh_wgrad_reduce"))


def _split_table(lib, call):
    """lh_wgrad_table_run = the table grid of the weight-gradient kernel, then the fold grid: the same call with
    lh_wgrad_table_info.run_parts = 1 / 2 runs each by itself (None when the table needs no fold launch)."""
    import copy
    import ctypes as C
    from lighthand_amd._lib import check
    blob, info_ref, dt = call.args
    info = info_ref._obj
    if info.n_fold_items <= 0:
        return None
    a, b = copy.copy(info), copy.copy(info)
    a.run_parts, b.run_parts = 1, 2
    call.keep = (call.keep, a, b)
    return (lambda st: check(lib.lh_wgrad_table_run(blob, C.byref(a), dt, st), "lh_wgrad_table_run (gradient grid)"),
            lambda st: check(lib.lh_wgrad_table_run(blob, C.byref(b), dt, st), "lh_wgrad_table_run (fold grid)"))


def epe_auc_parity():
    """The "EPE/AUC parity" half of BASELINE.json's metric: pred_eval's per-category AUC and EPE (src/utils/argparser.py:326-388)
    computed by the DEVICE path (lh_pck_curve -> metrics.auc_from_counts) on the synthetic evaluation set of golden G7, against the
    values the REFERENCE's own pred_eval produced for that set (tests/golden/g7_metrics.json, generated by importing the reference:
    tests/golden/make_golden.py) -- thresholds pckb [0.1, 0.3], mm [0, 30], mm [0, 50], four occlusion categories.
    The model-side parity (heat-maps, arg-max indices) lives in the tests; this is the metric arithmetic on identical predictions."""
    from lighthand_amd.metrics import auc_from_counts, device_pck_curve
    path = os.path.join(ROOT, "tests", "golden", "g7_metrics.json")
    try:
        g = json.load(open(path))
    except (OSError, ValueError):
        return {"status": "tests/golden/g7_metrics.json missing"}
    cats = g["evaluation"][0]
    d_auc = d_epe = 0.0
    n = 0
    curves_equal = True
    for key, T, method in (("pckb", [0.1, 0.3], "pckb"), ("mm30", [0, 30], "mm"), ("mm50", [0, 50], "mm")):
        for cat, v in cats.items():
            pred = torch.tensor(v["pred"], dtype=torch.float32).cuda()
            gt = torch.tensor(v["gt"], dtype=torch.float32).cuda()
            bb = torch.tensor(v["bb"], dtype=torch.float32).cuda()
            counts, nvis, dsum, nall = (t.cpu().numpy() for t in device_pck_curve(pred, gt, bb, T, method))
            auc, epe, curve = auc_from_counts(counts, nvis[0], dsum[0], nall[0], T, method)
            want = g["pred_eval"][key][cat]
            d_auc, d_epe = max(d_auc, abs(auc - want[0])), max(d_epe, abs(epe - want[1]))
            curves_equal = curves_equal and bool(np.allclose(curve, want[2], rtol=0, atol=1e-9))
            n += 1
    return {"max_abs_auc_diff": d_auc, "max_abs_epe_mm_diff": d_epe, "pck_curves_equal": curves_equal, "cases": n,
            "against": "the reference's pred_eval on the same predictions (golden G7: 4 occlusion categories x pckb[0.1,0.3] / mm[0,30] / mm[0,50])",
            "within_0p05_mm": bool(d_epe < 0.05)}


def kernel_roofline(flops, nbytes, ms, es=2):
    """a launch's lower bounds on both roofs, the governing one, and the measured time against them."""
    peak_tf = PEAK_BF16_TFLOPS if es == 2 else PEAK_F32_TFLOPS
    t_mfma, t_hbm = flops / (peak_tf * 1e12) * 1e3, nbytes / (PEAK_HBM_GBS * 1e9) * 1e3
    hbm = t_hbm >= t_mfma
    ach = (nbytes / (ms * 1e-3) / 1e9) if hbm else (flops / (ms * 1e-3) / 1e12)
    return {"bound": "hbm" if hbm else "mfma", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS if hbm else peak_tf,
            "unit": "GB/s" if hbm else "TFLOP/s", "frac": round(max(t_mfma, t_hbm) / ms, 4),
            "mfma_frac": round(t_mfma / ms, 4), "hbm_frac": round(t_hbm / ms, 4)}


def profile_average_ns(stats_csv, kernel):
    """AverageNs of `kernel` (bench.py's demangled name) in a rocprofv3 --stats kernel table, or None."""
    import csv
    import re
    types = {"DF16b": "__bf16", "DF16_": "_Float16", "f": "float"}
    try:
        rows = list(csv.DictReader(open(stats_csv)))
    except OSError:
        return None
    for r in rows:
        nm = r.get("Name", "")
        m = re.match(r"_Z\d+([A-Za-z_0-9]+?)I(DF16b|DF16_|f)((?:Li\d+E)*)E", nm)
        if m:
            ints = re.findall(r"Li(\d+)E", m.group(3))
            nm = m.group(1) + "<" + ", ".join([types[m.group(2)]] + ints) + ">"
        if nm == kernel or nm.replace(" ", "") == kernel.replace(" ", ""):
            return float(r["AverageNs"])
    return None


def cpu_baseline(depth, size, batch, seconds_budget=25.0, cores=None):
    """The oracle (plain PyTorch fp32 on the host cores) running the same training step on a
    bounded sample of the workload."""
    from oracle import heatmap as oh
    from oracle import models as omod
    import types
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    ns = types.SimpleNamespace
    extra = ns(NUM_LAYERS=depth, DECONV_WITH_BIAS=False, NUM_DECONV_LAYERS=3, NUM_DECONV_FILTERS=[256] * 3,
               NUM_DECONV_KERNELS=[4] * 3, FINAL_CONV_KERNEL=1)
    torch.manual_seed(9001)
    sd = omod.clone_state(get_pose_net(ns(MODEL=ns(EXTRA=extra, STYLE="pytorch")), True).state_dict())
    cores = cores or min(16, len(os.sched_getaffinity(0)))   # the GPU box gives one GPU a 16-core CPU share
    torch.set_num_threads(cores)
    rng = np.random.RandomState(9001)
    x = torch.from_numpy(rng.randn(batch, 3, size, size).astype(np.float32))
    joints = rng.uniform(20, size - 20, size=(batch, 21, 2)).astype(np.float32)
    adam = omod.AdamState(lr=1e-3)
    fwd = lambda s, xx: omod.pose_resnet_forward(s, xx, depth, "pytorch", training=True)

    def one():
        tgt = torch.from_numpy(np.stack([oh.generate_target(j) for j in joints]))[:, :, :size // 4, :size // 4]
        loss, pred, grads = omod.loss_and_grads(sd, fwd, x, tgt)
        oh.get_max_preds(pred.numpy())
        adam.step(sd, grads)

    one()                                                       # warm-up
    t0, n = time.time(), 0
    while True:
        one()
        n += 1
        if time.time() - t0 > seconds_budget * 0.6 or n >= (3 if batch >= 16 else 10):
            break
    dt = (time.time() - t0) / n
    return dict(value=round(batch / dt, 2), unit="images/s", cores=cores, kind="port",
                sample=f"oracle (plain PyTorch fp32 CPU) R{depth} {size}x{size} train step, batch {batch}, {n} timed step(s) of {dt:.2f} s")


def spawn_ranks(n):
    """Run this script under torch.distributed.run with n ranks on this node (one per GPU) and pass its output through.
    Returns the launcher's exit code; non-zero too when no rank printed the JSON line or the line reports another rank count."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    ranks_seen = None
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.startswith("{") and '"metric"' in line:
            try:
                ranks_seen = json.loads(line).get("n_gpus")
            except ValueError:
                pass
    rc = proc.wait()
    if rc == 0 and ranks_seen != n:
        print(f"bench.py: --gpus {n} but the run reported n_gpus = {ranks_seen}", file=sys.stderr)
        return 3
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--hrnet-width", type=int, default=0, help="benchmark HRNet-W<width> instead of SimpleBaseline (configs[3])")
    ap.add_argument("--infer-only", action="store_true", help="inference graph only (configs[4]: --size 384 --batch 256 --precision fp16)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--comm", default="torch", choices=["torch", "lh"],
                    help="gradient transport for --gpus > 1: torch.distributed all-reduce between per-segment graphs (default) or "
                         "the C-ABI communicator lh_comm_* (RCCL inside ONE captured graph per step)")
    ap.add_argument("--grad-buckets", default="fp32", choices=["fp32", "bf16"], help="dtype the gradient buckets travel in")
    ap.add_argument("--grad-algo", default="allreduce", choices=["allreduce", "direct"],
                    help="how a gradient bucket is exchanged: RCCL's all-reduce (its own algorithm choice), or 'direct' = all-to-all + local "
                         "sum + all-gather, i.e. reduce-scatter and all-gather with all seven xGMI peers at once (SURVEY 8e).  With --comm lh the three "
                         "steps are C-ABI launches inside the ONE captured graph of the step")
    ap.add_argument("--bucket-mib", type=int, default=64,
                    help="gradient bucket size for --gpus > 1.  64 MiB = 3 segments for R50: the data-parallel FORM of the step costs "
                         "+0.29 ms on one GPU (32 MiB / 5 segments: +0.63 ms; profiles/r05_dp_bucket_schedule_r50.txt), at the price of a "
                         "longer exposed tail if RCCL's ring stays at one link's rate (DESIGN.md section 6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra configurations (HRNet-W32 training, R50 384x384 fp16 inference)")
    ap.add_argument("--train-only", action="store_true",
                    help="training replays only (no inference graph, no per-launch profile, no extra configurations): the process the "
                         "PMC passes of tools/refresh_profiles.sh count a training step's HBM bytes on")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this parent (which has made no GPU call yet) starts the N ranks
        # as a CHILD launcher process, relays rank 0's JSON line and returns the launcher's exit code.  A worker never
        # re-executes itself.
        raise SystemExit(spawn_ranks(args.gpus))

    from lighthand_amd import parallel
    from lighthand_amd.runtime import InferStep, TrainStep

    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks_seen = 1
    if world > 1:                                            # the rank count the collective itself sees
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        ranks_seen = int(one.item())
    model = build_model(args.depth, args.precision, args.hrnet_width)
    name = f"HRNet-W{args.hrnet_width}" if args.hrnet_width else f"SimpleBaseline-ResNet{args.depth}"
    if args.infer_only:
        images, _ = synthetic_batch(args.batch, args.size, dev, seed=9001 + rank)
        model.eval()
        inf = InferStep(model, args.batch, args.size, args.size)
        inf.images.copy_(images)
        for _ in range(args.warmup):
            inf()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            inf()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        key = (f"hrnet{args.hrnet_width}" if args.hrnet_width else f"r{args.depth}", args.size)
        print(json.dumps({"metric": METRIC, "value": round(args.batch * args.steps / dt, 1), "unit": "images/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                          "config": {"workload": f"{name} {args.size}x{args.size} inference (eval-mode forward, BN folded, + argmax decode), "
                                                 f"batch {args.batch}, hipGraph replay"},
                          "step_roofline": step_roofline(key, False, args.batch * args.steps / dt, 4 if args.precision == "fp32" else 2)}))
        return
    sync = None
    if world > 1:
        sync = parallel.GradSync(world, bucket_bytes=args.bucket_mib << 20, compress="bf16" if args.grad_buckets == "bf16" else None,
                                 comm=parallel.LhComm() if args.comm == "lh" else None, algo=args.grad_algo)
    step = TrainStep(model, args.batch, args.size, args.size, lr=1e-3, use_graph=not args.no_graph, grad_sync=sync)
    images, joints = synthetic_batch(args.batch, args.size, dev, seed=9001 + rank)
    step.images.copy_(images)
    step.joints.copy_(joints)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # the K timed steps (wall clock between barriers = the contract's number); every step is also bracketed by HIP events on
    # the launch stream for the per-step median SURVEY 8d asks for
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(step.loss)
    ms = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed
    # what the gradient exchange costs the step: the same ranks, the same step, with the collectives stubbed out
    # (GradSync.stub: stream hand-over, staging and the per-bucket Adam launches stay).  Default transport only -- the
    # single-graph transport (--comm lh) holds its collectives inside the captured graph.
    exposed = None
    if world > 1 and sync is not None and args.comm == "torch":
        def timed(k):
            barrier()
            t_ = time.perf_counter()
            for _ in range(k):
                step()
            barrier()
            dt_ = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(dt_, op=torch.distributed.ReduceOp.MAX)
            return float(dt_) / k * 1e3
        k2 = max(2, min(args.steps, 30))
        with_ms = timed(k2)
        # with the collectives stubbed every rank applies its OWN gradients: weights and Adam state diverge across the ranks.
        # Snapshot them first and put them back afterwards, so the ranks are bit-identical again for whatever follows.
        arena_ = model.arena()
        flat_ = step.optimizer.state.get("flat", {})
        snap = [arena_.flat.clone()] + [flat_[k].clone() for k in ("exp_avg", "exp_avg_sq") if k in flat_]
        step_count = step.optimizer._dev[0]["step"].clone() if 0 in step.optimizer._dev else None
        bufs_ = {k: v.clone() for k, v in model.named_buffers()}
        sync.stub = True
        for _ in range(2):
            step()
        stub_ms = timed(k2)
        sync.stub = False
        torch.cuda.synchronize()
        arena_.flat.copy_(snap[0])
        for k, t_ in zip([k for k in ("exp_avg", "exp_avg_sq") if k in flat_], snap[1:]):
            flat_[k].copy_(t_)
        if step_count is not None:
            step.optimizer._dev[0]["step"].copy_(step_count)
        for k, v in model.named_buffers():
            v.copy_(bufs_[k])
        exposed = {"allreduce_exposed_ms": round(with_ms - stub_ms, 3), "ms_per_step_with_collectives": round(with_ms, 3),
                   "ms_per_step_collectives_stubbed": round(stub_ms, 3), "steps": k2}

    out = {
        "metric": METRIC, "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"{name} {args.size}x{args.size} training step "
                               f"(fwd + JointsMSELoss + argmax decode + bwd + Adam), batch {args.batch}/GPU, 21 joints, "
                               f"random init (seed 9001), hipGraph replay" + (f", dp{world} RCCL all-reduce ({args.comm}, {args.grad_algo}, {args.grad_buckets} buckets of {args.bucket_mib} MiB)" if world > 1 else ""),
                   "global_batch": world * args.batch, "parallelism": f"dp{world}"},
        "ms_per_step_median": round(median_ms, 3), "ranks_seen": ranks_seen,
        "dist_backend": torch.distributed.get_backend() if world > 1 else None,
        "loss_after": round(loss_val, 6),
        "allreduce_exposed_ms": exposed["allreduce_exposed_ms"] if exposed else None, "allreduce_exposed": exposed,
        "train_tflops": round(value * (TRAIN_GFLOP_PER_IMG if (args.depth, args.size) == (50, 256) else 0) / 1e3, 1),
    }
    wkey = (f"hrnet{args.hrnet_width}" if args.hrnet_width else f"r{args.depth}", args.size)
    es = 4 if args.precision == "fp32" else 2
    # whole-step roofline: per-GPU rate against the per-GPU peaks (2.760 TFLOP / 12.06 GB per R50 bs64 step)
    out["step_roofline"] = step_roofline(wkey, True, value / world, es)
    out["c_abi_calls_per_step"] = sum(1 for c in step.plan.packs + step.plan.fwd + step.plan.bwd if hasattr(c, "fn")) + 4

    if rank == 0 and not args.train_only:
        out["epe_auc_parity"] = epe_auc_parity()
    if rank == 0 and world == 1 and not args.train_only:
        # eval-mode forward + decode throughput (the "infer" half of the metric)
        torch.cuda.synchronize()
        inf = InferStep(model, args.batch, args.size, args.size)
        inf.images.copy_(images)
        for _ in range(3):
            inf()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            inf()
        torch.cuda.synchronize()
        out["infer_images_per_s"] = round(args.batch * args.steps / (time.perf_counter() - t1), 1)
        out["infer_step_roofline"] = step_roofline(wkey, False, out["infer_images_per_s"], es)

        if not args.no_roofline:
            agg = profile_kernels(step)
            # The process also replays the inference graph; fold its launches in with the counts this run
            # executed, so that per-kernel averages are comparable with `rocprofv3 --stats` of this command.
            agg_inf = profile_kernels(step, plan=inf.plan, fwd_only=True)
            n_train = args.warmup + args.steps + 1 + 4          # + capture warm-up + the 4 profiling passes
            n_inf = 3 + args.steps + 1 + 4
            mix = {}
            for src, n in ((agg, n_train), (agg_inf, n_inf)):
                for k, v in src.items():
                    m = mix.setdefault(k, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
                    for f in ("ms", "flops", "bytes"):
                        m[f] += v[f] * n
                    m["launches"] += v["launches"] * n
            # dominant SINGLE kernel (entries that aggregate several kernels of one C-ABI call are reported as a FAMILY below:
            # they cannot be matched to one rocprof row)
            name, _ = max(((k, v) for k, v in agg.items() if "(all kernels)" not in k), key=lambda kv: kv[1]["ms"])
            d = mix[name]
            tot = sum(v["ms"] for v in mix.values())
            avg_ms = d["ms"] / d["launches"]
            es_ = 2 if args.precision != "fp32" else 4
            # governing roof per launch = max(T_mfma, T_hbm) of its algorithmic FLOPs and bytes; both fractions reported
            out["roofline"] = kernel_roofline(d["flops"] / d["launches"], d["bytes"] / d["launches"], avg_ms, es_)
            out["roofline"]["traffic"] = None
            # HBM bytes per launch of that kernel from the committed PMC passes of this same command (profiles/README.md;
            # tools/pmc_traffic.py applies the guide's KiB unit and gfx950 FETCH_SIZE x2 correction).  A committed figure is
            # attached only while the profile still describes the kernels of THIS build: the profile's average duration of the
            # kernel (rocprofv3 --kernel-trace --stats of the same command) must lie within 10 % of the live HIP-event average.
            headline = args.depth == 50 and args.batch == 64 and args.size == 256 and args.precision == "bf16"
            prof_dir = os.path.join(ROOT, "profiles")
            pmc = {}
            try:
                pmc = json.load(open(os.path.join(prof_dir, PMC_FILE)))
            except (OSError, ValueError):
                out["roofline"]["traffic_source"] = "none: profiles/%s is missing or unreadable" % PMC_FILE
            if pmc and not headline:
                out["roofline"]["traffic_source"] = "none: the committed PMC passes describe the headline configuration only"
            elif pmc and name not in pmc:
                out["roofline"]["traffic_source"] = "none: profiles/%s has no row for this kernel (the dominant kernel changed since the profile)" % PMC_FILE
            elif pmc:
                prof_ns = profile_average_ns(os.path.join(prof_dir, STATS_FILE), name)
                if prof_ns is None:
                    out["roofline"]["traffic_source"] = "none: profiles/%s has no row for this kernel" % STATS_FILE
                elif abs(prof_ns * 1e-6 - avg_ms) > 0.10 * avg_ms:
                    out["roofline"]["traffic_source"] = ("none: the committed profile no longer describes this kernel (AverageNs %.0f in profiles/%s "
                                                         "vs %.0f ns live: more than 10 %% apart)" % (prof_ns, STATS_FILE, avg_ms * 1e6))
                else:
                    out["roofline"]["traffic"] = pmc[name]["read_bytes_per_launch"] + pmc[name]["write_bytes_per_launch"]
                    out["roofline"]["profile_avg_launch_ms"] = round(prof_ns * 1e-6, 4)
                    out["roofline"]["traffic_source"] = ("committed profile (profiles/%s; its AverageNs for this kernel is within 10 %% of the live "
                                                         "average), not measured in this run" % PMC_FILE)
                    out["roofline"]["traffic_note"] = ("bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, corrected per the "
                                                       "MI355X guide), profiles/%s" % PMC_FILE.replace("traffic.json", "hbm_traffic.txt"))
            # the waste ratio of the WHOLE training step: fabric bytes (PMC read + write over one step of a --train-only process)
            # over SURVEY 8(d)'s algorithmic bytes
            st = pmc.get("__train_step__") if headline else None
            alg = WORK[("r50", 256)][3] * 1e6 * args.batch
            out["roofline"]["step_traffic"] = ({"bytes": int(st["bytes"]), "read_bytes": int(st["read_bytes"]), "write_bytes": int(st["write_bytes"]),
                                                "algorithmic_bytes": int(alg), "over_algorithmic": round(st["bytes"] / alg, 3),
                                                "source": "committed profile (profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                          "`bench.py --train-only`, all kernels of %d training steps / %d), not measured in this run"
                                                          % (PMC_FILE, st["steps"], st["steps"])} if st else
                                               {"bytes": None, "over_algorithmic": None, "algorithmic_bytes": int(alg),
                                                "source": "none: profiles/%s has no __train_step__ entry for this configuration" % PMC_FILE})
            out["roofline"].update({"kernel": name, "launches_per_train_step": agg[name]["launches"], "avg_launch_ms": round(avg_ms, 4),
                                    "flops_per_launch": int(d["flops"] / d["launches"]), "algorithmic_bytes_per_launch": int(d["bytes"] / d["launches"]),
                                    "share_of_profiled_ms": round(d["ms"] / tot, 3),
                                    "event_pair_overhead_us": round(agg[name]["gap_ms"] / max(agg[name]["launches"], 1) * 1e3, 2),
                                    "note": "average over every launch of this kernel in the process (train-step and inference-graph launches, "
                                            "weighted by how often each ran), HIP events on the launch stream minus half the measured empty-pair "
                                            "time: comparable with the AverageNs of rocprofv3 --kernel-trace --stats (profiles/" + STATS_FILE + ")"})
            # the largest FAMILY of the step: the BatchNorm / ReLU backward (reduce + coefficient fold + apply kernels per call)
            fam = agg.get("fuse_bwd(all kernels)")
            if fam and fam["ms"] > 0:
                fr = kernel_roofline(0.0, fam["bytes"], fam["ms"], es_)
                fr.update({"family": "fuse_bwd (fuse_bwd_reduce* + fuse_bwd_coef_fused + fuse_bwd_apply*)", "calls_per_train_step": fam["launches"],
                           "ms_per_train_step": round(fam["ms"], 3), "algorithmic_bytes_per_step": int(fam["bytes"]),
                           "share_of_profiled_ms": round(fam["ms"] / sum(v["ms"] for v in agg.values()), 3)})
                # `frac` above is against the family's OWN pass count (five tensor passes per BatchNorm node, three behind a gated data gradient), NOT a roofline fraction
                # in SURVEY 8(d)'s sense: 8(d) charges the BatchNorm backward one re-read of y per BatchNorm term, everything
                # else these kernels move is traffic a perfect fusion would not have
                b8 = getattr(step.plan, "bn_bwd_8d_bytes", 0.0)
                if b8 > 0:
                    fr["section_8d_bytes_per_step"] = int(b8)
                    fr["over_section_8d"] = round(fam["bytes"] / b8, 2)
                    fr["frac_of_section_8d_roof"] = round(b8 / (PEAK_HBM_GBS * 1e9) * 1e3 / fam["ms"], 4)
                    fr["note"] = ("frac = the family's own algorithmic bytes (5 tensor passes per BatchNorm node; 3 where a gated data gradient, "
                                  "lh_igemm_gated, did the reduce pass in its epilogue -- its read of x is charged to that launch) over the HBM peak; "
                                  "frac_of_section_8d_roof = the bytes SURVEY 8(d) charges to it (one re-read of y) over the HBM peak")
                out["roofline_family"] = fr
            # per kernel: [ms per step, launches, TFLOP/s of its algorithmic FLOPs, GB/s of its algorithmic bytes, governing roof of the
            # sum of its launches] -- the stage-1 / stage-2 kernels that look slow in TFLOP/s are the HBM-bound ones
            out["kernel_breakdown_ms"] = {k: [round(v["ms"], 3), v["launches"],
                                              round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                                              round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 0) if v["bytes"] and v["ms"] > 0 else None,
                                              ("hbm" if v["bytes"] / (PEAK_HBM_GBS * 1e9) >= v["flops"] / ((PEAK_F32_TFLOPS if es_ == 4 else PEAK_BF16_TFLOPS) * 1e12)
                                               else "mfma") if (v["bytes"] or v["flops"]) else None]
                                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        if not args.no_extra and (args.depth, args.size, args.batch, args.hrnet_width) == (50, 256, 64, 0):
            # the other single-GPU configurations of BASELINE.json, timed the same way (hipGraph replays, inputs resident)
            extra = {}
            del inf
            # the same inference with TWO batches in flight (runtime.InferPipeline: one captured graph per slot, a stream each):
            # the serving form -- infer_images_per_s above stays the one-batch-in-flight number of the earlier rounds
            from lighthand_amd.runtime import InferPipeline
            pipe = InferPipeline(model, args.batch, args.size, args.size, depth=2)
            for st_ in pipe.steps:
                st_.images.copy_(images)
            for _ in range(6):
                pipe.submit()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(2 * args.steps):
                pipe.submit()
            torch.cuda.synchronize()
            extra["r50_infer_256_bs64_two_in_flight"] = {"images_per_s": round(args.batch * 2 * args.steps / (time.perf_counter() - t2), 1),
                                                         "in_flight": 2, "dtype": args.precision}
            del pipe
            # configs[3], one GPU's share.  Timed dtype: fp16 with the static loss scale (TrainStep default 1024) -- BASELINE.json
            # names no dtype for this configuration, and HRNet's heat-maps are 8x closer to the fp32 oracle in fp16 than in
            # bf16 at the same speed (tests/test_gpu_runtime.py::test_c4_...); bf16 is reported beside it
            for key4, prec4 in (("hrnet_w32_train_bs32", "fp16"), ("hrnet_w32_train_bs32_bf16", "bf16")):
                m4 = build_model(precision=prec4, hrnet_width=32)
                s4 = TrainStep(m4, 32, 256, 256, lr=1e-3)
                im4, j4 = synthetic_batch(32, 256, dev)
                s4.images.copy_(im4); s4.joints.copy_(j4)
                dt4 = timed_replays(s4, 5, 20)
                extra[key4] = {
                    "images_per_s": round(32 / dt4, 1), "ms_per_step": round(dt4 * 1e3, 3), "dtype": prec4, "loss_scale": s4.loss_scale,
                    "c_abi_calls_per_step": sum(1 for c in s4.plan.packs + s4.plan.fwd + s4.plan.bwd if hasattr(c, "fn")) + 4,
                    "step_roofline": step_roofline(("hrnet32", 256), True, 32 / dt4, 2)}
                del s4, m4
            # the headline workload in the REFERENCE's own arithmetic (fp32 everywhere, src/utils/method.py:160-183 has no
            # autocast): the plan the 1e-3 parity claim lives on, priced against the fp32-matrix peak (157 TFLOP/s)
            m32 = build_model(args.depth, "fp32")
            s32 = TrainStep(m32, args.batch, args.size, args.size, lr=1e-3)
            s32.images.copy_(images); s32.joints.copy_(joints)
            dt32 = timed_replays(s32, 3, 10)
            extra["r50_train_256_bs64_fp32"] = {
                "images_per_s": round(args.batch / dt32, 1), "ms_per_step": round(dt32 * 1e3, 3), "dtype": "fp32",
                "step_roofline": step_roofline(("r50", 256), True, args.batch / dt32, 4)}
            del s32, m32
            m5 = build_model(50, "fp16").eval()                                             # configs[4]
            i5 = InferStep(m5, 256, 384, 384)
            i5.images.copy_(synthetic_batch(256, 384, dev)[0])
            dt5 = timed_replays(i5, 3, 10)
            extra["r50_infer_384_bs256_fp16"] = {"images_per_s": round(256 / dt5, 1), "ms_per_step": round(dt5 * 1e3, 3), "dtype": "fp16",
                                                 "step_roofline": step_roofline(("r50", 384), False, 256 / dt5, 2)}
            del i5, m5
            out["extra"] = extra
        if not args.no_cpu_baseline:
            # the oracle on the host cores: the headline workload on a bounded sample, and BASELINE.json configs[0]
            # (R18, 256x256, bs 8) at 8 threads (comparable with BASELINE.md section 2) and at the box's core share
            out["cpu_baseline"] = cpu_baseline(args.depth, args.size, batch=16)
            ncore = min(16, len(os.sched_getaffinity(0)))
            out["cpu_baseline_c1"] = [cpu_baseline(18, 256, batch=8, seconds_budget=12.0, cores=c) for c in sorted({min(8, ncore), ncore})]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
