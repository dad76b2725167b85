"""GPU: the hipGraph fast path (TrainStep / InferStep), the data-parallel gradient path emulated on one
GPU, checkpoints in the reference's format, and the train CLI end to end."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import resnet_cfg

pytestmark = pytest.mark.gpu


def _model(depth=18, precision="fp32", seed=9001):
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    torch.manual_seed(seed)
    return get_pose_net(resnet_cfg(depth), True).cuda().set_precision(precision)


def _batch(b, size, seed):
    rng = np.random.RandomState(seed)
    return (torch.from_numpy(rng.randn(b, 3, size, size).astype(np.float32)).cuda(),
            torch.from_numpy(rng.uniform(8, size - 8, size=(b, 21, 2)).astype(np.float32)).cuda())


def test_graph_step_equals_eager_step_and_dropin_loop():
    """Three iterations: (a) hipGraph TrainStep, (b) eager TrainStep, (c) the reference-style loop through the
    drop-in API (model(), JointsMSELoss, backward(), Adam.step()) -- same losses, same weights."""
    from lighthand_amd.heatmap import JointsMSELoss, render_targets
    from lighthand_amd.optim import Adam
    from lighthand_amd.runtime import TrainStep
    x, j = _batch(4, 64, 1)
    results = []
    for mode in ("graph", "eager", "dropin"):
        m = _model()
        losses = []
        if mode == "dropin":
            opt = Adam(m.parameters(), lr=1e-3).bind_arena(m.arena())
            crit = JointsMSELoss(False)
            for _ in range(3):
                loss = crit(m(x), render_targets(j)[:, :, :16, :16].contiguous(), None)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        else:
            step = TrainStep(m, 4, 64, 64, lr=1e-3, use_graph=(mode == "graph"))
            for _ in range(3):
                losses.append(float(step(x, j)))
            assert step.preds.shape == (4, 21, 2) and float(step.preds.max()) <= 63 * 4
        results.append((losses, m.arena().flat.clone(), {k: v.clone() for k, v in m.named_buffers()}))
    (la, wa, ba), (lb, wb, bb), (lc, wc, bc) = results
    assert np.allclose(la, lb, rtol=1e-6) and np.allclose(la, lc, rtol=1e-5), (la, lb, lc)
    assert torch.allclose(wa, wb, rtol=1e-5, atol=1e-7) and torch.allclose(wa, wc, rtol=1e-4, atol=1e-6)
    for k in ba:
        assert torch.allclose(ba[k].float(), bb[k].float(), rtol=1e-5, atol=1e-7), k
    assert int(ba["bn1.num_batches_tracked"]) == 3          # warm-up / capture iterations are rolled back


def test_data_parallel_semantics_emulated_on_one_gpu(monkeypatch):
    """World of 2 emulated: rank 0 runs TrainStep whose GradSync adds 'rank 1's' gradients bucket by bucket
    (as the RCCL all-reduce would); the result must equal averaging the two per-rank gradients (each with its
    own BatchNorm statistics) followed by one Adam step -- SURVEY.md section 8(e) oracle."""
    from lighthand_amd import parallel
    from lighthand_amd.heatmap import JointsMSELoss, render_targets
    from lighthand_amd.optim import Adam
    from lighthand_amd.runtime import TrainStep
    xa, ja = _batch(4, 64, 11)
    xb, jb = _batch(4, 64, 12)
    crit = JointsMSELoss(False)

    def grads_of(x, j):
        m = _model()
        loss = crit(m(x), render_targets(j)[:, :, :16, :16].contiguous(), None)
        loss.backward()
        return m, m.arena().flat_grad.clone()

    m_ref, ga = grads_of(xa, ja)
    _, gb = grads_of(xb, jb)
    opt = Adam(m_ref.parameters(), lr=1e-3).bind_arena(m_ref.arena())
    m_ref.arena().flat_grad.copy_((ga + gb) / 2)
    opt.step()
    want = m_ref.arena().flat.clone()

    class PeerSync(parallel.GradSync):
        def __init__(self, peer_grad):
            super().__init__(world_size=2, bucket_bytes=2 << 20)
            self.peer, self.launched = peer_grad, []

        def launch(self, flat_grad, bucket, after=None):
            s, e = bucket
            flat_grad[s:e] += self.peer[s:e]           # what all_reduce(SUM) over 2 ranks leaves behind
            self.launched.append(bucket)
            if after is not None:                      # the bucket's Adam update rides behind its all-reduce
                after(torch.cuda.current_stream().cuda_stream)

    for use_graph, group in ((False, None), (True, None), (True, "4"), (True, "0")):
        if group is None:
            monkeypatch.delenv("LH_WGRAD_GROUP", raising=False)
        else:
            monkeypatch.setenv("LH_WGRAD_GROUP", group)     # finer deferred groups (more cuts) / weight gradients in place
        m = _model()
        sync = PeerSync(gb)
        step = TrainStep(m, 4, 64, 64, lr=1e-3, use_graph=use_graph, grad_sync=sync)
        step(xa, ja)
        torch.cuda.synchronize()
        got = m.arena().flat
        # weight gradients finish in deferred groups (16 layers each by default): R18's 21 convolutions give two cuts
        assert len(sync.segments(step.plan)) >= (2 if group is None else 3), "R18 at 2 MiB buckets must be cut into several segments"
        covered = sorted(b for _, _, b in sync.segments(step.plan) if b)
        assert covered[0][0] == 0 and covered[-1][1] == m.arena().numel
        assert torch.allclose(got, want, rtol=2e-4, atol=2e-6), float((got - want).abs().max())


def test_infer_step_matches_eval_forward_and_oracle_decode():
    from lighthand_amd.runtime import InferStep
    from oracle.heatmap import get_max_preds
    m = _model(18)
    x, _ = _batch(4, 64, 3)
    m.train()
    with torch.no_grad():
        m(x)                                                  # give the running statistics some content
    m.eval()
    with torch.no_grad():
        y = m(x).cpu().numpy()
    inf = InferStep(m, 4, 64, 64)
    preds = inf(x).cpu().numpy()
    inf(x)                                                    # graph replay
    assert np.allclose(inf.heatmaps.cpu().numpy(), y, rtol=1e-5, atol=1e-6)
    assert np.array_equal(inf.preds.cpu().numpy(), get_max_preds(y)[0] * 4)
    assert np.array_equal(preds, inf.preds.cpu().numpy())


def test_checkpoint_format_and_torch_adam_interop(tmp_path):
    """checkpoint-good/state_dict.bin keeps the reference's keys (src/tools/dataset.py:352-360); the optimizer
    entry loads into torch.optim.Adam and back."""
    import types
    from lighthand_amd.optim import Adam
    from lighthand_amd.runtime import TrainStep
    from lighthand_amd.tools import train as T
    m = _model(18)
    x, j = _batch(2, 64, 5)
    step = TrainStep(m, 2, 64, 64, lr=1e-3, use_graph=False)
    for _ in range(2):
        step(x, j)
    args = types.SimpleNamespace(output_dir=str(tmp_path / "output" / "simplebaseline" / "ours" / "84k"))
    T.save_checkpoint(m, args, epoch=4, optimizer=step.optimizer, best_loss=0.5, count=1)
    path = os.path.join(args.output_dir, "checkpoint-good", "state_dict.bin")
    sd = torch.load(path, map_location="cpu")
    assert sorted(sd) == ["best_loss", "count", "epoch", "model_state_dict", "optimizer_state_dict"]
    assert len(sd["model_state_dict"]) == 122 + 0 * len(sd) or len(sd["model_state_dict"]) == len(m.state_dict())
    # torch.optim.Adam accepts the optimizer state (reference loop: optimizer.load_state_dict, train.py:50)
    cpu_model = copy.deepcopy(m).cpu()
    t_opt = torch.optim.Adam(cpu_model.parameters(), lr=1e-3)
    t_opt.load_state_dict(sd["optimizer_state_dict"])
    st0 = t_opt.state[next(iter(cpu_model.parameters()))]
    assert float(st0["step"]) == 2 and st0["exp_avg"].abs().sum() > 0
    # and back: resume into a fresh model + fused Adam
    m2 = _model(18, seed=1)
    best, epo, count, opt_state = T.resume_checkpoint(m2, path)
    assert (best, epo, count) == (0.5, 5, 1)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    o2 = Adam(m2.parameters(), lr=1e-3).bind_arena(m2.arena())
    o2.load_state_dict(opt_state)
    assert torch.equal(o2.state["flat"]["exp_avg"], step.optimizer.state["flat"]["exp_avg"])
    assert int(o2._dev[0]["step"]) == 2


def test_train_cli_end_to_end(tmp_path, capsys):
    from lighthand_amd.tools import train as T
    args = T.parse_args(["--root_path", str(tmp_path), "--synthetic", "32", "--val_synthetic", "8", "--batch_size", "8",
                         "--epoch", "3", "--depth", "18", "--size", "64", "--precision", "bf16", "--reset"])
    best = T.main(args)
    out = capsys.readouterr().out
    assert "valid loss" in out and np.isfinite(best)
    assert os.path.isfile(os.path.join(args.output_dir, "checkpoint-good", "state_dict.bin"))
    first = float(out.split("valid loss ")[1].split()[0])
    assert best <= first


def test_train_cli_trains_and_validates_on_the_short_last_batch(tmp_path, capsys):
    """src/tools/train.py:27-38 builds both loaders with the default drop_last=False: 20 training samples at batch 8 are THREE iterations per
    epoch (8, 8, 4) and 12 validation samples two batches (8, 4).  The short batch runs on a step of its own shape that shares the optimizer:
    after two epochs Adam has taken 6 steps (the checkpoint's optimizer state says so), and --drop_last gives the 4-step run."""
    from lighthand_amd.tools import train as T
    steps = {}
    for tag, extra in (("ref", []), ("drop", ["--drop_last"])):
        args = T.parse_args(["--root_path", str(tmp_path / tag), "--synthetic", "20", "--val_synthetic", "12", "--batch_size", "8", "--epoch", "2",
                             "--depth", "18", "--size", "64", "--precision", "bf16", "--reset", "--count", "5"] + extra)
        best = T.main(args)
        out = capsys.readouterr().out
        assert np.isfinite(best) and out.count("valid loss") == 2
        assert ("iter 0/3" in out) == (tag == "ref") and ("iter 0/2" in out) == (tag == "drop"), out
        sd = torch.load(os.path.join(args.output_dir, "checkpoint-good", "state_dict.bin"), map_location="cpu")
        st = sd["optimizer_state_dict"]["state"]
        steps[tag] = (int(float(st[0]["step"])), sd["epoch"])
    assert steps["ref"][0] == 3 * (steps["ref"][1] + 1) and steps["drop"][0] == 2 * (steps["drop"][1] + 1), steps


def test_train_entry_point_takes_datasets(tmp_path, capsys, monkeypatch):
    """main(args, train_set=, val_set=) (reference: src/tools/train.py:24-38 builds the datasets): any Dataset of
    (image, joint_2d, ...) samples.  (a) the reference's sample layout -- normalised float [3, S, S], joints [21, 3],
    an extra heat-map the engine re-renders itself; (b) RAW uint8 HWC frames through the fused device pipeline with the
    colour jitter on the reference's FIXED subset idx < len * ratio_of_aug (src/tools/dataset.py:133), uint8 validation set."""
    from lighthand_amd import runtime
    from lighthand_amd.tools import train as T

    class RefLayout(torch.utils.data.Dataset):                    # (image, joint_2d[21, 3], heatmap) like CustomDataset
        def __init__(self, n, seed):
            rng = np.random.RandomState(seed)
            self.x = rng.randn(n, 3, 64, 64).astype(np.float32)
            self.j = np.concatenate([rng.uniform(8, 56, size=(n, 21, 2)), np.ones((n, 21, 1))], -1).astype(np.float32)

        def __len__(self):
            return len(self.x)

        def __getitem__(self, i):
            return torch.from_numpy(self.x[i]), torch.from_numpy(self.j[i]), torch.zeros(21, 16, 16)

    class RawFrames(torch.utils.data.Dataset):                    # uint8 [H, W, 3] camera frames, joints already in the 64-px frame
        def __init__(self, n, seed):
            rng = np.random.RandomState(seed)
            self.x = rng.randint(0, 256, size=(n, 48, 80, 3)).astype(np.uint8)
            self.j = rng.uniform(8, 56, size=(n, 21, 2)).astype(np.float32)

        def __len__(self):
            return len(self.x)

        def __getitem__(self, i):
            return self.x[i], self.j[i]

    common = ["--root_path", str(tmp_path), "--batch_size", "8", "--epoch", "2", "--depth", "18", "--size", "64", "--precision", "bf16", "--reset"]
    # loaders without worker processes (the reference uses eight, src/utils/pre_argparser.py): every epoch forks the workers of
    # both loaders from a process that holds a HIP context and hundreds of GB of mappings -- ~20 s per fork, 180 of this
    # suite's 435 s with even one worker per loader
    args0 = T.parse_args(common + ["--name", "ref"])
    args0.num_workers = 0
    best = T.main(args0, train_set=RefLayout(32, 1), val_set=RefLayout(8, 2))
    assert np.isfinite(best) and "valid loss" in capsys.readouterr().out
    seen = []
    real = runtime.sample_color_jitter
    monkeypatch.setattr(runtime, "sample_color_jitter", lambda n, *a, mask=None, **k: (seen.append(mask.clone()), real(n, *a, mask=mask, **k))[1])
    args = T.parse_args(common + ["--name", "raw", "--ratio_of_aug", "0.25"])
    args.num_workers = 0
    best = T.main(args, train_set=RawFrames(32, 3), val_set=RawFrames(8, 4))
    assert np.isfinite(best)
    assert len(seen) == 2 * 4 and all(m.dtype == torch.bool and m.numel() == 8 for m in seen)
    assert sum(int(m.sum()) for m in seen) == 2 * 8               # idx < 32 * 0.25: the same 8 samples, once per epoch
    assert os.path.isfile(os.path.join(args.output_dir, "checkpoint-good", "state_dict.bin"))
    try:
        T.main(T.parse_args(common))                              # neither datasets nor --synthetic
    except SystemExit as e:
        assert "train_set" in str(e)
    else:
        raise AssertionError("main() without data must exit")


def test_eval_cli_writes_reference_formats(tmp_path):
    """train -> checkpoint-good/state_dict.bin -> wearable_eval_2d: evaluation.json + three pck_eval_*.txt files whose
    numbers equal the oracle's pred_eval on the stored predictions."""
    import json
    from lighthand_amd.tools import train as T
    from lighthand_amd.tools import wearable_eval_2d as E
    from oracle import metrics as om
    args = T.parse_args(["--root_path", str(tmp_path), "--root", "simplebaseline/frei", "--name", "run1", "--synthetic", "16",
                         "--val_synthetic", "8", "--batch_size", "8", "--epoch", "1", "--depth", "18", "--size", "64",
                         "--precision", "fp32", "--reset"])
    T.main(args)
    files = E.main(["--root_path", str(tmp_path), "--model_path", "simplebaseline/frei", "--batch_size", "8", "--depth", "18",
                    "--size", "64", "--synthetic", "20"])
    assert len(files) == 3 and all(os.path.isfile(f) for f in files)
    ev = json.load(open(os.path.join(str(tmp_path), "simplebaseline/frei/run1", "evaluation.json")))
    assert isinstance(ev, list) and set(ev[0]) == set(E.CATEGORIES)
    n = sum(len(v["bb"]) for v in ev[0].values())
    assert n == 20 and all(len(p) == 21 for v in ev[0].values() for p in v["pred"])
    want = om.pred_eval({k: v for k, v in ev[0].items() if v["bb"]}, [0.1, 0.3], "pckb")
    line = [l for l in open(files[0]) if l.startswith("mean_auc;")][0].split(";")
    assert line[1] == "simplebaseline/frei/run1"
    assert abs(float(line[2]) - want["mean_auc"][0]) < 0.006 and abs(float(line[3]) - want["mean_auc"][1]) < 0.006
    # the device-side reduction (lh_pck_curve; one host read at the end) reproduces the AUC of all visible joints
    torch.manual_seed(9001)
    model = T.build_model(args).cuda()
    model.load_state_dict(torch.load(os.path.join(str(tmp_path), "simplebaseline/frei/run1/checkpoint-good/state_dict.bin"),
                                     map_location="cpu")["model_state_dict"], strict=False)
    loader = torch.utils.data.DataLoader(E.SyntheticEvalSet(20, 64), batch_size=8, shuffle=False)
    dev = E.device_eval(model.train(), loader, 8, 64, bn_train=True)
    for ty, T_list in E.THRESHOLDS:
        host = om.pred_eval({k: v for k, v in ev[0].items() if v["bb"]}, T_list, ty)["mean_auc"]
        assert abs(dev[(ty, T_list[1])][0] - host[0]) < 1e-6, (ty, T_list)


def test_resume_in_graph_mode_keeps_adam_state(tmp_path):
    """optimizer.load_state_dict() BEFORE the first captured step (src/tools/train.py:50): the warm-up / capture
    iterations of TrainStep must hand the loaded moments and step count back, so that k steps + save + resume + 1 step
    equals k + 1 uninterrupted steps."""
    import types
    from lighthand_amd.optim import Adam
    from lighthand_amd.runtime import TrainStep
    from lighthand_amd.tools import train as T
    x, j = _batch(4, 64, 31)
    k = 3
    m_ref = _model(18)
    ref = TrainStep(m_ref, 4, 64, 64, lr=1e-3, use_graph=True)
    for _ in range(k + 1):
        ref(x, j)
    m_a = _model(18)
    a = TrainStep(m_a, 4, 64, 64, lr=1e-3, use_graph=True)
    for _ in range(k):
        a(x, j)
    args = types.SimpleNamespace(output_dir=str(tmp_path / "run"))
    T.save_checkpoint(m_a, args, epoch=0, optimizer=a.optimizer, best_loss=1.0, count=0)
    m_b = _model(18, seed=5)                                   # different init: everything must come from the file
    _, _, _, opt_state = T.resume_checkpoint(m_b, os.path.join(args.output_dir, "checkpoint-good", "state_dict.bin"))
    opt = Adam(m_b.parameters(), lr=1e-3)
    b = TrainStep(m_b, 4, 64, 64, optimizer=opt, use_graph=True)
    opt.load_state_dict(opt_state)                              # same order as tools/train.py main()
    b(x, j)
    torch.cuda.synchronize()
    assert int(opt._dev[0]["step"]) == k + 1
    assert torch.allclose(m_b.arena().flat, m_ref.arena().flat, rtol=1e-5, atol=1e-7), \
        float((m_b.arena().flat - m_ref.arena().flat).abs().max())
    assert torch.allclose(opt.state["flat"]["exp_avg"], ref.optimizer.state["flat"]["exp_avg"], rtol=1e-4, atol=1e-9)
    for (kk, va), (_, vb) in zip(m_ref.named_buffers(), m_b.named_buffers()):
        assert torch.allclose(va.float(), vb.float(), rtol=1e-5, atol=1e-7), kk


@pytest.fixture
def static_kernel_choice(monkeypatch):
    """The tests below compare networks whose weights come out of >= 1000 16-bit training steps INSIDE the test.  With
    measured kernel choices (timing near-ties fall differently per process) two processes train different weights and the
    statistics move from run to run; with the library's static default configurations (LH_AUTOTUNE=0: no measurement, the
    weight gradient's split counts and every tile fixed by the launch shape alone) the whole trajectory is bit-reproducible
    across processes (test_static_choice_training_is_bit_reproducible_across_processes), and the bounds are the measured
    values plus a stated margin."""
    monkeypatch.setenv("LH_AUTOTUNE", "0")


_REPRO_SNIPPET = """
import hashlib, sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/tests')
from conftest import resnet_cfg
from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
from lighthand_amd.modeling.hrnet.pose_hrnet import get_hrnet, hrnet_cfg
from lighthand_amd.runtime import TrainStep
out = []
for tag in ('r50', 'hrnet32'):
    torch.manual_seed(9001)
    m = (get_pose_net(resnet_cfg(50), True) if tag == 'r50' else get_hrnet(hrnet_cfg(32), True)).cuda().set_precision('bf16')
    rng = np.random.RandomState(3)
    x = torch.from_numpy(rng.randn(4, 3, 128, 128).astype(np.float32)).cuda()
    j = torch.from_numpy(rng.uniform(10, 118, size=(4, 21, 2)).astype(np.float32)).cuda()
    step = TrainStep(m, 4, 128, 128, lr=1e-3)
    for _ in range(25):
        step(x, j)
    torch.cuda.synchronize()
    out.append(hashlib.sha256(m.arena().flat.cpu().numpy().tobytes()).hexdigest())
print('SHA ' + ' '.join(out))
"""


def test_sliced_adam_is_bit_identical_to_one_update(static_kernel_choice, monkeypatch):
    """LH_ADAM_SLICES=1: the arena is updated slice by slice while the backward pass still runs (lh_adam_tick once, then
    lh_adam_apply per finished gradient bucket on a side stream; data parallel: behind each bucket's all-reduce).  Adam is
    elementwise: weights and moments after three steps equal those of the single lh_adam_step launch bit for bit."""
    from lighthand_amd import parallel
    from lighthand_amd.runtime import TrainStep
    x, j = _batch(4, 64, 5)

    def run(sliced, sync):
        monkeypatch.setenv("LH_ADAM_SLICES", "1" if sliced else "0")
        monkeypatch.setenv("LH_ADAM_SLICE_MB", "2")
        m = _model(seed=77)
        step = TrainStep(m, 4, 64, 64, lr=1e-3, grad_sync=parallel.GradSync(world_size=1, bucket_bytes=2 << 20) if sync else None)
        assert step.adam_slices == sliced
        for _ in range(3):
            loss = step(x, j)
        torch.cuda.synchronize()
        if sliced and not sync:
            assert len(step._adam_segs) >= 2, step._adam_segs      # weight gradients finish in deferred groups: R18 gives two cuts
        st = step.optimizer.state["flat"]
        return float(loss), m.arena().flat.clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), int(step.optimizer._dev[0]["step"])

    want = run(False, False)
    for sync in (False, True):
        got = run(True, sync)
        assert got[0] == want[0] and got[4] == want[4] == 3
        for a, b in zip(got[1:4], want[1:4]):
            assert torch.equal(a, b)


def test_static_choice_training_is_bit_reproducible_across_processes(static_kernel_choice):
    """Two PROCESSES, same seed, LH_AUTOTUNE=0: 25 captured bf16 training steps of R50 and of HRNet-W32 (merged
    multi-problem launches, deferred weight gradients on side streams, split-K folds) end in bit-identical weights.
    This is what lets the trained-weight parity tests below carry bounds that do not move from run to run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _REPRO_SNIPPET.format(root=root)
    env = dict(os.environ, LH_AUTOTUNE="0")
    shas = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        shas.append([l for l in r.stdout.splitlines() if l.startswith("SHA ")][-1])
    assert shas[0] == shas[1], shas


def test_c5_r50_fp16_infer_384_graph_matches_oracle(static_kernel_choice):
    """BASELINE.json config 5 (R50 inference, 384 x 384, fp16, hipGraph replay; batch cut to what the CPU oracle finishes
    in seconds): eval-mode BN-folded InferStep vs oracle.models.pose_resnet_forward(training=False) on the same weights.
    Weights come from 2400 bf16 training steps on these (learnable, synthetic) images, so the network emits heatmap-like
    peaks -- the arg-max of a random-init network's flat maps is decided by rounding noise -- and its running
    statistics are non-trivial.  Declared fp16 tolerance: heatmaps within 2e-2 of the fp32 oracle's peak value
    (SURVEY section 7.2-F), arg-max keypoints equal on >= 99 % of the 210 joints (measured 100 %; the bound leaves room for
    two near-ties, the weights being the outcome of a chaotic training run), and bit-equal to the oracle's decode rule
    applied to the HIP heatmaps."""
    from lighthand_amd.runtime import InferStep, TrainStep
    from oracle import models as omod
    from oracle.heatmap import get_max_preds
    b, size = 10, 384
    m = _model(50, precision="bf16")
    # learnable synthetic "hands": one bright blob per image, the 21 joints at fixed offsets from it (a translation-
    # equivariant task a convolutional network memorises in ~1000 steps; on pure noise it stays at the all-zero plateau)
    rng = np.random.RandomState(21)
    cen = rng.uniform(100, size - 100, size=(b, 1, 2)).astype(np.float32)
    joints = cen + rng.uniform(-70, 70, size=(1, 21, 2)).astype(np.float32)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    img = 0.1 * rng.randn(b, 3, size, size).astype(np.float32)
    for i in range(b):
        blob = np.exp(-((xx - cen[i, 0, 0]) ** 2 + (yy - cen[i, 0, 1]) ** 2) / (2 * 14.0 ** 2))
        img[i] += 3.0 * blob[None] * np.array([1.0, 0.6, -0.8], np.float32)[:, None, None]
    x, j = torch.from_numpy(img).cuda(), torch.from_numpy(joints).cuda()
    step = TrainStep(m, b, size, size, lr=1e-3)
    for _ in range(2400):                                       # sharp peaks, converged running statistics
        step(x, j)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        want = omod.pose_resnet_forward(sd, x.cpu(), 50, training=False).numpy()
    m.eval().set_precision("fp16")
    inf = InferStep(m, b, size, size)
    inf(x)
    preds = inf(x).cpu().numpy()                                # second call = graph replay
    got = inf.heatmaps.cpu().numpy()
    err = float(np.abs(got - want).max() / np.abs(want).max())
    dec_h = get_max_preds(got)[0] * 4
    dec_o = get_max_preds(want)[0] * 4
    match = float((dec_h == dec_o).all(-1).mean())
    print(f"C5 parity: heatmap max err / peak {err:.3e}, peak {np.abs(want).max():.3f}, arg-max agreement {match:.4f}")
    assert got.shape == want.shape == (b, 21, 96, 96)
    assert np.abs(want).max() > 0.5                             # the network did learn peaks
    # measured (round 4, static kernel choice, identical in two processes): err 2.14e-3, arg-max 210 / 210
    assert err < 5e-3
    assert np.array_equal(preds, dec_h)                         # device decode == oracle rule on the same heatmaps
    assert match >= 0.995                                       # at most one of the 210 joints


def test_c2_r50_bf16_train_forward_matches_fp32_oracle(static_kernel_choice):
    """BASELINE.json config 2 -- the TIMED configuration (R50, 256 x 256, bf16; batch cut to 8 so the CPU oracle finishes
    in seconds) -- pinned by VALUE to the fp32 oracle: the train-mode (batch-statistics) forward and JointsMSELoss of the
    bf16 HIP path vs oracle.models on the same weights.  Weights come from 2400 bf16 training steps on learnable synthetic
    images, so the heat-maps carry real peaks (the arg-max of a random-init network's flat maps is decided by rounding
    noise) and the BN statistics are non-trivial.  Declared bf16 tolerances: heat-maps within
    5e-2 of the oracle's peak value at the WORST of the 688k elements (measured 1.7e-2 .. 3.6e-2 over runs: a tail
    statistic of 8-bit activations through 53 train-mode BN layers; SURVEY section 7.2-F's 2e-2 is met by the fp16 path of
    C5 at 1.8e-3) and 2e-3 in RMS (measured 6e-4), arg-max keypoints equal on >= 98 % of the 168 joints (measured 100 %; one joint off by a
    pixel in one run of many -- the weights under test come out of 2400 chaotic bf16 steps and depend on the process's kernel choices), loss within 5e-2
    relative (at convergence the loss is the small residual of two nearly equal maps: measured 3e-2).  The gradients of
    the same configuration are pinned on well-conditioned weights by
    test_gpu_model.py::test_c2_r50_bf16_gradients_vs_fp32_oracle."""
    from lighthand_amd.heatmap import JointsMSELoss, render_targets
    from lighthand_amd.runtime import TrainStep
    from oracle import models as omod
    from oracle.heatmap import get_max_preds
    b, size = 8, 256
    m = _model(50, precision="bf16")
    rng = np.random.RandomState(33)
    cen = rng.uniform(70, size - 70, size=(b, 1, 2)).astype(np.float32)
    joints = cen + rng.uniform(-48, 48, size=(1, 21, 2)).astype(np.float32)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    img = 0.1 * rng.randn(b, 3, size, size).astype(np.float32)
    for i in range(b):
        blob = np.exp(-((xx - cen[i, 0, 0]) ** 2 + (yy - cen[i, 0, 1]) ** 2) / (2 * 10.0 ** 2))
        img[i] += 3.0 * blob[None] * np.array([1.0, 0.6, -0.8], np.float32)[:, None, None]
    x, j = torch.from_numpy(img).cuda(), torch.from_numpy(joints).cuda()
    step = TrainStep(m, b, size, size, lr=1e-3)
    for _ in range(2400):
        step(x, j)
    torch.cuda.synchronize()
    sd = omod.clone_state({k: v.detach().cpu() for k, v in m.state_dict().items()})
    tgt = render_targets(j).cpu()
    with torch.no_grad():
        want = omod.pose_resnet_forward(sd, x.cpu(), 50, training=True).numpy()
    loss_ref = float(0.5 * ((want - tgt.numpy()) ** 2).mean())
    m.train()
    with torch.no_grad():
        pred = m(x)
        loss = JointsMSELoss(False)(pred, tgt.cuda(), None)
    got = pred.cpu().numpy()
    peak = float(np.abs(want).max())
    err = float(np.abs(got - want).max() / peak)
    rms = float(np.sqrt(((got - want) ** 2).mean()) / peak)
    match = float((get_max_preds(got)[0] == get_max_preds(want)[0]).all(-1).mean())
    lrel = abs(float(loss) - loss_ref) / abs(loss_ref)
    print(f"C2 bf16 forward parity: heatmap max err / peak {err:.3e}, rms / peak {rms:.3e} (peak {peak:.3f}), arg-max agreement {match:.4f}, "
          f"loss rel {lrel:.3e}")
    assert got.shape == want.shape == (b, 21, 64, 64)
    assert peak > 0.5                                           # the network did learn peaks
    # measured (round 4, static kernel choice, identical in two processes): max 3.78e-2, RMS 6.7e-4, arg-max 168 / 168, loss 2.7e-2
    assert err < 5e-2 and rms < 1e-3
    assert match >= 0.99                                        # at most one of the 168 joints
    assert lrel < 4e-2


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_c4_hrnet_w32_bf16_train_forward_matches_fp32_oracle(precision, static_kernel_choice):
    """BASELINE.json config 4's model in its timed dtype (HRNet-W32, 256 x 256, bf16; batch cut to 8 for the CPU oracle),
    pinned by VALUE to the fp32 oracle like C2 above: train-mode forward + JointsMSELoss of the bf16 HIP path (merged
    multi-problem launches, mixed BN-backward grids during the training that produces the weights) vs oracle.models
    (pose_hrnet.py:425-460) on the same weights after 1500 bf16 training steps on learnable synthetic images.  HRNet is
    harder on 8-bit mantissas than R50: the activations' relative RMS error grows from 4e-3 after the stem to 1.3e-2 at
    the last 64 x 64 feature map (no single operation stands out; the fp16 path of the same plan is 8x closer: RMS 1.2e-3,
    arg-max 98.8 %; the fp32 path agrees with the oracle to 2e-5), and the head's flat-topped peaks let that noise move
    an arg-max by a pixel.  The weights under test are themselves the product of 1 500 chaotic bf16 steps (they depend on
    the kernel choices of the process), so the statistics move from run to run.  Declared bf16 tolerances (DESIGN.md section
    4): heat-maps within 2e-2 of the oracle's peak in RMS (measured 0.7e-2 .. 1.1e-2 over runs) and 2.5e-1 at the 99.9th
    percentile (5.5e-2 .. 1.5e-1), arg-max key points within ONE heat-map pixel on >= 90 % of the joints (93-97 %; exactly
    equal on 81-89 %), loss within 1e-1 relative (2e-2 .. 6e-2); the worst single element is reported, not bounded
    (0.30-0.58 of the peak).  The fp16 case trains the same plan with its static loss scale (TrainStep default 1024) and is
    held to tighter bounds: RMS 4e-3 (measured 1.2e-3), 99.9th percentile 5e-2, within one pixel on >= 97 % of the joints."""
    from lighthand_amd.heatmap import JointsMSELoss, render_targets
    from lighthand_amd.modeling.hrnet.pose_hrnet import get_hrnet, hrnet_cfg
    from lighthand_amd.runtime import TrainStep
    from oracle import models as omod
    from oracle.heatmap import get_max_preds
    b, size = 8, 256
    torch.manual_seed(9001)
    m = get_hrnet(hrnet_cfg(32), True).cuda().set_precision(precision)
    rng = np.random.RandomState(33)
    cen = rng.uniform(70, size - 70, size=(b, 1, 2)).astype(np.float32)
    joints = cen + rng.uniform(-48, 48, size=(1, 21, 2)).astype(np.float32)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    img = 0.1 * rng.randn(b, 3, size, size).astype(np.float32)
    for i in range(b):
        blob = np.exp(-((xx - cen[i, 0, 0]) ** 2 + (yy - cen[i, 0, 1]) ** 2) / (2 * 10.0 ** 2))
        img[i] += 3.0 * blob[None] * np.array([1.0, 0.6, -0.8], np.float32)[:, None, None]
    x, j = torch.from_numpy(img).cuda(), torch.from_numpy(joints).cuda()
    step = TrainStep(m, b, size, size, lr=1e-3)
    assert step.plan.batch and step.plan._n_groups > 0          # the merged-launch plan is the one that trains
    assert step.loss_scale == (1024.0 if precision == "fp16" else 1.0)
    for _ in range(1500):
        step(x, j)
    torch.cuda.synchronize()
    sd = omod.clone_state({k: v.detach().cpu() for k, v in m.state_dict().items()})
    tgt = render_targets(j).cpu()
    with torch.no_grad():
        want = omod.hrnet_forward(sd, x.cpu(), training=True).numpy()
    loss_ref = float(0.5 * ((want - tgt.numpy()) ** 2).mean())
    m.train()
    with torch.no_grad():
        pred = m(x)
        loss = JointsMSELoss(False)(pred, tgt.cuda(), None)
    got = pred.cpu().numpy()
    peak = float(np.abs(want).max())
    err = float(np.abs(got - want).max() / peak)
    rms = float(np.sqrt(((got - want) ** 2).mean()) / peak)
    pg, pw = get_max_preds(got)[0], get_max_preds(want)[0]
    match = float((pg == pw).all(-1).mean())
    near = float((np.abs(pg - pw).max(-1) <= 1).mean())
    p999 = float(np.quantile(np.abs(got - want), 0.999) / peak)
    lrel = abs(float(loss) - loss_ref) / abs(loss_ref)
    print(f"C4 (HRNet-W32) {precision} forward parity: heatmap rms / peak {rms:.3e}, 99.9th percentile {p999:.3e}, max {err:.3e} (peak {peak:.3f}), "
          f"arg-max equal {match:.4f}, within one pixel {near:.4f}, loss rel {lrel:.3e}")
    assert got.shape == want.shape == (b, 21, 64, 64)
    assert peak > 0.5                                           # the network did learn peaks
    # measured (round 4, static kernel choice, identical in two processes) -- bounds = measured value + a margin for
    # legitimate changes of rounding (a different summation order moves the 1 500-step trajectory):
    #   fp16: RMS 1.5e-3, 99.9th percentile 2.0e-2, worst element 7.4e-2, equal 98.2 %, within one pixel 100 %, loss 1.4e-3
    #   bf16: RMS 9.3e-3, 99.9th percentile 1.09e-1, worst element 5.6e-1, equal 90.5 %, within one pixel 96.4 %, loss 2.9e-2
    #   round 6 (table launches of the weight gradient: other pixel-split counts, i.e. another fp32 summation order of dW -- the gradients
    #   agree with the per-layer launches to 2e-5, tests/test_gpu_wgrad_table.py -- and therefore another 1 500-step bf16 trajectory):
    #   bf16: RMS 9.6e-3, 99.9th 1.17e-1, worst 5.5e-1, equal 85.1 % (143 of 168 joints), within one pixel 95.8 %, loss 3.4e-2; fp16 unchanged
    #   inside its bounds.  The bf16 arg-max share of THIS network moves by several joints between equally valid roundings (DESIGN.md
    #   section 4: fp16 + static loss scale is C4's timed dtype for that reason); its bound is 80 % since round 6, the others stand.
    if precision == "fp16":
        assert rms < 2.5e-3 and p999 < 3e-2 and err < 1.2e-1
        assert near >= 0.99 and match >= 0.96
        assert lrel < 5e-3
    else:
        assert rms < 1.3e-2 and p999 < 1.5e-1 and err < 7.5e-1
        assert near >= 0.94 and match >= 0.80
        assert lrel < 5e-2


def test_eval_tail_batch_runs_unpadded(tmp_path):
    """N % batch != 0 in the reference-quirk mode (pred_store runs train-mode BN, argparser.py:246-281): the short
    last batch is normalised with ITS OWN batch statistics, as in the reference loop -- checked against the oracle's
    train-mode forward of exactly those samples."""
    from lighthand_amd.tools import wearable_eval_2d as E
    from oracle import models as omod
    from oracle.heatmap import get_max_preds
    m = _model(18)
    data = E.SyntheticEvalSet(20, 64)
    loader = torch.utils.data.DataLoader(data, batch_size=8, shuffle=False)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    meta = E.pred_store(m.train(), loader, str(tmp_path / "evaluation.json"), 8, 64, bn_train=True)
    assert sum(len(v["bb"]) for v in meta.values()) == 20
    got = {c: iter(v["pred"]) for c, v in meta.items()}
    agree = total = 0
    sd = {k: v.clone() for k, v in sd0.items()}
    for images, _, cats in loader:                              # batches of 8, 8, 4 through the oracle, train mode
        with torch.no_grad():
            hm = omod.pose_resnet_forward(sd, images, 18, training=True).numpy()
        want = get_max_preds(hm)[0] * 4
        for i, c in enumerate(cats):
            p = np.asarray(next(got[c]), np.float32)
            agree += int((p == want[i]).all(-1).sum())
            total += 21
    print("tail-batch eval: arg-max agreement with the oracle", agree / total)
    assert agree / total >= 0.99


def _dp_rank(rank, world, port, compress, q, own_gpu=False, lh_comm=False):
    """One data-parallel rank (spawned fresh: the parent's GPU context is never re-used or re-exec'ed).  own_gpu: one GPU
    per rank and RCCL (multi-GPU nodes); otherwise both ranks share GPU 0 and the collective goes through gloo."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank) if own_gpu else "0", LH_DIST_BACKEND="nccl" if own_gpu else "gloo")
    import torch.distributed as dist
    from lighthand_amd import parallel
    from lighthand_amd.runtime import TrainStep
    parallel.init_distributed()
    m = _model(18)
    x, j = _batch(4, 64, 11 + rank)
    sync = parallel.GradSync(world, bucket_bytes=2 << 20, compress=compress, comm=parallel.LhComm() if lh_comm else None)
    step = TrainStep(m, 4, 64, 64, lr=1e-3, use_graph=True, grad_sync=sync)
    losses = [float(step(x, j)) for _ in range(2)]
    torch.cuda.synchronize()
    segs = sync.segments(step.plan)
    q.put((rank, m.arena().flat.cpu().numpy(), losses, len(segs), [b for _, _, b in segs]))   # by value: the rank may exit first
    dist.barrier()
    step.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("compress", [None, "bf16"])
def test_two_process_data_parallel_step_with_real_collective(compress, monkeypatch, tmp_path):
    """The per-segment hipGraph path with a REAL collective between two processes (gloo through the host, both ranks on
    this GPU; on a multi-GPU node the same code runs over RCCL): after two steps both ranks hold identical weights,
    equal to the single-process emulation -- two micro-batches with their own BatchNorm statistics, gradients averaged,
    one Adam step each (SURVEY 8e oracle)."""
    import socket
    import torch.multiprocessing as mp
    from lighthand_amd.heatmap import JointsMSELoss, render_targets
    from lighthand_amd.optim import Adam
    # MEASURED kernel configurations, identical everywhere: rank 0 tunes and broadcasts its choices to rank 1
    # (parallel.plan_with_shared_tuning) and persists them (LH_TUNE_CACHE, default on); the emulation below -- another
    # process, as a resumed job would be -- starts from that file.  (Per-process tuning could pick different pixel-split
    # counts on a timing near-tie: a different fp32 summation order, which Adam's first steps amplify to +-lr.)
    from lighthand_amd.engine import Plan
    monkeypatch.setenv("LH_TUNE_CACHE", str(tmp_path / "tune.txt"))
    monkeypatch.setenv("LH_TUNE_DB", "0")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_rank, args=(r, 2, port, compress, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=300)
        res[r[0]] = (r[0], torch.from_numpy(r[1]),) + tuple(r[2:])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0][1], res[1][1]), "ranks diverged"
    assert res[0][3] >= 2 and res[0][4] == res[1][4]                 # several segments, same buckets on both ranks
    assert os.path.getsize(str(tmp_path / "tune.txt")) > 0           # rank 0 measured and persisted its choices
    monkeypatch.setattr(Plan, "_tune_file_loaded", False)            # this process now starts from rank 0's file
    # single-process emulation of the two ranks
    crit = JointsMSELoss(False)
    m = _model(18)
    opt = Adam(m.parameters(), lr=1e-3).bind_arena(m.arena())
    batches = [_batch(4, 64, 11), _batch(4, 64, 12)]
    for _ in range(2):            # (train-mode BN normalises with batch statistics: the running buffers do not enter the gradients)
        grads = []
        for x, j in batches:
            loss = crit(m(x), render_targets(j)[:, :, :16, :16].contiguous(), None)
            opt.zero_grad()
            loss.backward()
            grads.append(m.arena().flat_grad.clone())
        m.arena().flat_grad.copy_((grads[0] + grads[1]) / 2)
        opt.step()
    want = m.arena().flat.cpu()
    diff = (res[0][1] - want).abs()
    if compress is None:
        assert torch.allclose(res[0][1], want, rtol=2e-4, atol=2e-6), float(diff.max())
        # ... and pinned to the CPU ORACLE's emulation: oracle.models.loss_and_grads on each rank's micro-batch
        # (own BatchNorm statistics), gradients averaged, oracle AdamState.step -- nothing of the HIP engine on this side
        from oracle import models as omod
        sd = omod.clone_state({k: v.detach().cpu() for k, v in _model(18).state_dict().items()})
        adam = omod.AdamState(lr=1e-3)
        cpu_batches = [(x.cpu(), render_targets(j)[:, :, :16, :16].contiguous().cpu()) for x, j in batches]
        for _ in range(2):
            gs = [omod.loss_and_grads(sd, lambda s_, xx: omod.pose_resnet_forward(s_, xx, 18, training=True), x, t)[2] for x, t in cpu_batches]
            adam.step(sd, {k: (gs[0][k] + gs[1][k]) / 2 for k in gs[0]})
        offs = m.arena().offsets
        d = torch.cat([(res[0][1][o:o + n_] - sd[k].reshape(-1)).abs() for k, (o, n_, _) in offs.items()])
        # Adam's first steps move an element by ~lr * sign(g): where a gradient is within rounding of zero the two sides may
        # pick different signs, so the bound is per element 2 steps * lr (x2 for opposite directions) and the SHARE of such
        # elements must be small; everything else agrees to 1e-4
        share = float((d > 1e-4).float().mean())
        print(f"data-parallel step vs CPU-oracle emulation: max |dw| {float(d.max()):.3e}, share of elements off by > 1e-4: {share:.4f}")
        assert float(d.max()) <= 2 * 2 * 1e-3 * 1.05 and share < 0.02
    else:
        # bf16 buckets round every rank's gradient to 8 significant bits before the sum: where the two ranks' values
        # nearly cancel the sign of the sum can flip, and Adam's first steps move such an element by +-lr either way
        print("bf16 buckets: max |dw|", float(diff.max()), "share of elements off by > 1e-4:", float((diff > 1e-4).float().mean()))
        assert float(diff.max()) <= 2 * 2 * 1e-3 * 1.05 and float((diff > 1e-4).float().mean()) < 0.02


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL with two ranks)")
def test_lh_comm_two_gpus_single_graph_matches_torch_transport(monkeypatch, tmp_path):
    """Two ranks on two GPUs: the C-ABI communicator (lh_comm_*: RCCL launches inside ONE captured graph per step) and
    the default transport (torch.distributed all-reduce between per-segment graphs) give bit-identical weights, equal
    across ranks.  Skipped on one-GPU boxes; `bench.py --gpus N --comm lh` stays experimental until this has run."""
    import socket
    import torch.multiprocessing as mp
    monkeypatch.setenv("LH_TUNE_CACHE", str(tmp_path / "tune.txt"))
    out = {}
    for lh in (False, True):
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_dp_rank, args=(r, 2, port, None, q, True, lh)) for r in range(2)]
        for p in procs:
            p.start()
        res = {}
        for _ in procs:
            r = q.get(timeout=300)
            res[r[0]] = torch.from_numpy(r[1])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert torch.equal(res[0], res[1]), "ranks diverged"
        out[lh] = res[0]
    assert torch.equal(out[False], out[True])


def test_bf16_bucket_staging_c_abi():
    """lh_cast_f32_bf16 (the staging of GradSync(compress='bf16')): round-to-nearest-even like the framework's cast,
    NaN / inf preserved, any length and alignment; and back."""
    from lighthand_amd import _lib
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    for n, off in ((1 << 20, 0), (1003, 1), (7, 3)):
        src = torch.randn(n + 8, device="cuda")[off:off + n]
        src[:3] = torch.tensor([float("nan"), float("inf"), -float("inf")], device="cuda")[:min(3, n)]
        half = torch.empty(n + 8, dtype=torch.bfloat16, device="cuda")[off:off + n]
        _lib.check(lib.lh_cast_f32_bf16(src.data_ptr(), half.data_ptr(), n, 0, s))
        assert torch.equal(half.view(torch.int16), src.to(torch.bfloat16).view(torch.int16))
        back = torch.zeros_like(src)
        _lib.check(lib.lh_cast_f32_bf16(back.data_ptr(), half.data_ptr(), n, 1, s))
        assert torch.equal(back.view(torch.int32), half.float().view(torch.int32))


def test_lh_comm_c_abi_single_rank():
    """lh_comm_* (RCCL behind the C ABI) with a one-rank communicator -- what one GPU can exercise: id, init, an in-place
    all-reduce on a side stream (the sum over one rank is the identity), the same call captured in a hipGraph, destroy.
    Multi-rank behaviour is RCCL's; the bucketed use is covered by the gloo tests with the same GradSync code."""
    from lighthand_amd import parallel
    comm = parallel.LhComm(rank=0, world_size=1)
    x = torch.randn(1 << 20, device="cuda")
    want = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm.all_reduce_sum_(x, stream=side)
    for dt in (torch.bfloat16, torch.float16):
        h = want.to(dt)
        comm.all_reduce_sum_(h, stream=side)
        side.synchronize()
        assert torch.equal(h, want.to(dt))
    side.synchronize()
    assert torch.equal(x, want)
    # through GradSync (world 2 semantics need two GPUs; here: the plumbing, bucket views, bf16 staging)
    sync = parallel.GradSync(world_size=2, bucket_bytes=1 << 20, comm=comm, compress="bf16")
    flat = torch.randn(3 << 18, device="cuda")
    ref = flat.clone()
    sync.launch(flat, (0, 1 << 18))
    sync.wait_all()
    torch.cuda.synchronize()
    assert torch.equal(flat[1 << 18:], ref[1 << 18:]) and torch.equal(flat[:1 << 18], ref[:1 << 18].to(torch.bfloat16).float())
    # the data-parallel step with the collectives INSIDE one captured graph equals the eager segment-by-segment step
    from lighthand_amd.runtime import TrainStep
    xb, jb = _batch(4, 64, 41)
    got = []
    for use_graph in (True, False):
        m = _model(18)
        step = TrainStep(m, 4, 64, 64, lr=1e-3, use_graph=use_graph, grad_sync=parallel.GradSync(world_size=2, bucket_bytes=2 << 20, comm=comm))
        for _ in range(2):
            step(xb, jb)
        torch.cuda.synchronize()
        got.append(m.arena().flat.clone())
        if use_graph:
            assert len(step.graphs) == 1
    assert torch.allclose(got[0], got[1], rtol=1e-5, atol=1e-7), float((got[0] - got[1]).abs().max())
    # round 6: the pieces of the DIRECT exchange at the C ABI (lh_comm_alltoall / lh_sum_chunks / lh_comm_allgather, and RCCL's own
    # reduce-scatter): with one rank every one of them is the identity; ragged lengths go through the padded staging buffer; and the
    # direct exchange lives inside the single captured graph of the data-parallel step (GradSync(algo="direct", comm=...))
    import ctypes as C
    from lighthand_amd import _lib
    lib = _lib.load()
    rk, nr = C.c_int(-1), C.c_int(-1)
    assert lib.lh_comm_size(comm.handle, C.byref(rk), C.byref(nr)) == 0 and (rk.value, nr.value) == (0, 1)
    src = torch.randn(4099, device="cuda")
    for fn in (lib.lh_comm_reduce_scatter_sum, lib.lh_comm_allgather, lib.lh_comm_alltoall):
        dst = torch.full_like(src, float("nan"))
        assert fn(comm.handle, src.data_ptr(), dst.data_ptr(), src.numel(), _lib.LH_F32, torch.cuda.current_stream().cuda_stream) == 0, lib.lh_last_error()
        torch.cuda.synchronize()
        assert torch.equal(dst, src)
    assert lib.lh_comm_alltoall(comm.handle, src.data_ptr(), src.data_ptr(), 16, _lib.LH_F32, None) == -1      # not in place
    assert lib.lh_comm_allgather(comm.handle, src.data_ptr(), src.data_ptr(), 16, 7, None) == -1               # bad dtype
    for n in (1 << 16, 4099):
        for dt in (torch.float32, torch.bfloat16):
            t = torch.randn(n, device="cuda").to(dt)
            want = t.clone()
            comm.direct_sum_(t, stream=side)
            side.synchronize()
            assert torch.equal(t, want), (n, dt)
    got2 = []
    for use_graph in (True, False):
        m = _model(18)
        step = TrainStep(m, 4, 64, 64, lr=1e-3, use_graph=use_graph,
                         grad_sync=parallel.GradSync(world_size=2, bucket_bytes=2 << 20, comm=comm, algo="direct"))
        for _ in range(2):
            step(xb, jb)
        torch.cuda.synchronize()
        got2.append(m.arena().flat.clone())
        if use_graph:
            assert len(step.graphs) == 1
    assert torch.allclose(got2[0], got2[1], rtol=1e-5, atol=1e-7)
    assert torch.equal(got2[1], got[1])                        # one rank: the direct exchange and the all-reduce are both the identity
    comm.close()


def test_noop_device_move_keeps_a_live_step_and_a_real_move_invalidates_it():
    """model.cuda() on a model that already lives on the device leaves its arena (and the raw pointers a captured step
    holds) alone; a move that re-creates the parameter storages makes the stale step refuse to run instead of training
    buffers state_dict() no longer sees."""
    from lighthand_amd._lib import LightHandError
    from lighthand_amd.runtime import TrainStep
    m = _model(18)
    x, j = _batch(2, 64, 5)
    step = TrainStep(m, 2, 64, 64, lr=1e-3, use_graph=True)
    step(x, j)
    flat = m.arena().flat
    m.cuda()                                            # no-op: same storages
    assert m.arena().flat.data_ptr() == flat.data_ptr()
    before = flat.clone()
    step(x, j)
    torch.cuda.synchronize()
    assert not torch.equal(m.arena().flat, before)      # the live step still trains what state_dict() sees
    off, numel, _ = step.plan.arena_offsets["conv1.weight"]
    assert torch.equal(m.state_dict()["conv1.weight"].flatten(), m.arena().flat[off:off + numel])
    m.double().float()                                  # re-creates every storage
    with pytest.raises(LightHandError, match="re-created"):
        step(x, j)


def test_pred_store_test_file_feeds_pred_test(tmp_path):
    """The category-less evaluation file (argparser.py:284-323): one entry per batch, coordinates in the 256-pixel frame,
    and pred_test on it equals the oracle's pred_test on the same file."""
    import json
    from lighthand_amd import metrics as M
    from lighthand_amd.tools import wearable_eval_2d as E
    from oracle import metrics as om
    m = _model(18)
    data = E.SyntheticEvalSet(16, 64)
    loader = torch.utils.data.DataLoader(data, batch_size=8, shuffle=False)
    path = str(tmp_path / "final_model" / "fx" / "test.json")
    meta = E.pred_store_test(m.train(), loader, path, 8, 64)
    on_disk = json.load(open(path))
    assert isinstance(on_disk, list) and set(on_disk[0]) == {"pred", "gt", "bb"} and len(on_disk[0]["pred"]) == 2
    assert np.asarray(on_disk[0]["pred"][0]).shape == (8, 21, 2) and len(on_disk[0]["bb"][1]) == 8
    for T, method in (([0.1, 0.3], "pckb"), ([0, 30], "mm")):
        assert M.pred_test(meta, T, method) == om.pred_test(on_disk[0], T, method)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_fused_inference_head_matches_separate_launches(precision):
    """Inference plans run `final_layer(relu(bn(deconv(x))))` as ONE launch (lh_igemm_phases_head: the 256-channel
    activation never reaches HBM).  Against the same plan with the head as separate launches (Plan.fuse_head = False):
    two launches fewer, heat-maps equal up to the 16-bit rounding the separate path applies to its NHWC heat-map before
    the fp32 conversion (the fused path writes fp32 directly), decode of each path bit-equal to the oracle's decode."""
    from lighthand_amd.engine import Plan
    from lighthand_amd.runtime import InferStep
    from oracle.heatmap import get_max_preds
    x, _ = _batch(3, 96, 21)
    out = {}
    try:
        for fused in (True, False):
            Plan.fuse_head = fused
            m = _model(18, precision).eval()
            st = InferStep(m, 3, 96, 96)
            preds = st(x).clone()
            torch.cuda.synchronize()
            hm = st.heatmaps.clone()
            assert bool(getattr(st.plan, "_head_fused", False)) == fused
            want = get_max_preds(hm.cpu().numpy())[0] * 4
            assert np.array_equal(preds.cpu().numpy(), want)
            out[fused] = (hm, sum(1 for c in st.plan.fwd if hasattr(c, "fn")))
    finally:
        Plan.fuse_head = True
    assert out[True][1] == out[False][1] - 2
    scale = float(out[False][0].abs().max())
    assert float((out[True][0] - out[False][0]).abs().max()) <= scale * 2.0 ** (-8 if precision == "bf16" else -10)


def test_static_loss_scale_is_exact_and_defaults_to_1024_for_fp16():
    """TrainStep(loss_scale=S): the loss gradient is multiplied by S where it is formed and Adam divides it out again.  A
    power of two commutes with every rounding of the backward pass (no overflow / underflow at these magnitudes), so a bf16
    step with S = 64 leaves bit-identical weights and the same loss as S = 1; fp16 plans default to S = 1024 (their
    1e-7-sized heat-map gradients flush to zero otherwise: R50 loss after 100 steps 4.6e-3 unscaled vs 1.5e-3 scaled = bf16's),
    bf16 / fp32 plans to 1 (src/utils/method.py:160-183 is the loop this step replaces)."""
    from lighthand_amd.runtime import TrainStep
    x, j = _batch(4, 128, 5)
    res = []
    for scale in (1.0, 64.0):
        m = _model(18, precision="bf16")
        step = TrainStep(m, 4, 128, 128, lr=1e-3, loss_scale=scale)
        losses = []
        for _ in range(3):
            step(x, j)
            losses.append(float(step.loss))
        res.append((losses, m.arena().flat.clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])
    assert TrainStep(_model(18, precision="fp16"), 4, 128, 128).loss_scale == 1024.0
    assert TrainStep(_model(18, precision="bf16"), 4, 128, 128).loss_scale == 1.0


def test_infer_pipeline_two_batches_in_flight_equals_infer_step():
    """runtime.InferPipeline (several eval-mode batches in flight, a captured graph and a stream per slot; the decode of
    src/utils/argparser.py:246-281): every batch's key points and confidences equal InferStep's bit for bit, in submission
    order, for more batches than slots; a ticket that is no longer in flight is refused."""
    from lighthand_amd._lib import LightHandError
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    from lighthand_amd.runtime import InferPipeline, InferStep
    from conftest import resnet_cfg
    torch.manual_seed(3)
    model = get_pose_net(resnet_cfg(18), True).cuda().set_precision("bf16").eval()
    b, h, w = 4, 128, 96
    batches = [torch.randn(b, 3, h, w, device="cuda") for _ in range(5)]
    ref = InferStep(model, b, h, w, slot=7)
    want = []
    for x in batches:
        p = ref(x)
        torch.cuda.synchronize()
        want.append((p.clone(), ref.maxvals.clone()))
    pipe = InferPipeline(model, b, h, w, depth=2)
    got = list(pipe.map(batches))
    assert len(got) == len(want)
    for (p, m), (pw, mw) in zip(got, want):
        assert torch.equal(p, pw) and torch.equal(m, mw)
    with pytest.raises(LightHandError):
        pipe.result(0)
    # raw uint8 frames (ToTensor / Resize / Normalize fused on the device, dataset.py:128-159): same equality per slot
    frames = [torch.randint(0, 256, (b, 100, 80, 3), dtype=torch.uint8, device="cuda") for _ in range(3)]
    ref8 = InferStep(model, b, h, w, input_u8=(100, 80), slot=8)
    want8 = []
    for f in frames:
        p = ref8(f)
        torch.cuda.synchronize()
        want8.append(p.clone())
    pipe8 = InferPipeline(model, b, h, w, depth=2, input_u8=(100, 80))
    for (p, _), pw in zip(pipe8.map(frames), want8):
        assert torch.equal(p, pw)



def test_infer_steps_own_their_plans_and_pipeline_inputs_may_be_freed():
    """(1) An InferStep with uint8 input rewires ITS plan's image launch; ``model.eval()(x)`` for the same shape must still read
    x (the plan cache keys plans by owner: module.HipModule.plan).  (2) InferPipeline.submit copies the caller's batch on
    the slot's stream: a caller that drops the tensor right after submit() (what pipe.map over a generator does) may get the
    same block back from the caching allocator for the next batch -- the copy must have been recorded on the slot stream."""
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    from lighthand_amd.runtime import InferPipeline, InferStep
    torch.manual_seed(4)
    model = get_pose_net(resnet_cfg(18), True).cuda().set_precision("bf16").eval()
    b, h, w = 4, 128, 96
    x1, x2 = torch.randn(b, 3, h, w, device="cuda"), torch.randn(b, 3, h, w, device="cuda")
    with torch.no_grad():
        y1 = model(x1).clone()
    step8 = InferStep(model, b, h, w, input_u8=(100, 80))
    step8(torch.randint(0, 256, (b, 100, 80, 3), dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    with torch.no_grad():
        y1b, y2 = model(x1).clone(), model(x2).clone()
    assert torch.equal(y1, y1b), "model(x) must not be affected by an InferStep(input_u8=...) of the same shape"
    assert not torch.equal(y1, y2), "model(x) must read x"
    # (2) inputs freed and reallocated between submits
    gen = torch.Generator(device="cuda").manual_seed(7)
    ref = InferStep(model, b, h, w)
    want = []
    for i in range(6):
        x = torch.randn(b, 3, h, w, device="cuda", generator=gen)
        want.append(ref(x).clone())
        torch.cuda.synchronize()
    gen = torch.Generator(device="cuda").manual_seed(7)
    pipe = InferPipeline(model, b, h, w, depth=2)

    def batches():
        for i in range(6):
            yield torch.randn(b, 3, h, w, device="cuda", generator=gen)      # dropped by map() right after submit
    for (p, _), pw in zip(pipe.map(batches()), want):
        assert torch.equal(p, pw)


def test_dropped_infer_steps_release_their_plans():
    """Round-4 advisor: every InferStep got a plan of its own that the module cached for good, so a loop that builds a step
    per epoch / per batch size grew device memory without bound.  Owner plans are no longer cached: building N steps and
    dropping them leaves neither entries in model._lh_plans nor allocated memory behind."""
    import gc
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    from lighthand_amd.runtime import InferStep
    torch.manual_seed(5)
    model = get_pose_net(resnet_cfg(18), True).cuda().set_precision("bf16").eval()
    b, h, w = 4, 128, 96
    x = torch.randn(b, 3, h, w, device="cuda")

    def one():
        st = InferStep(model, b, h, w)
        out = st(x).clone()
        torch.cuda.synchronize()
        del st
        return out
    first = one()
    gc.collect()
    torch.cuda.synchronize()
    n_plans, mem = len(model._lh_plans), torch.cuda.memory_allocated()
    for _ in range(4):
        assert torch.equal(one(), first)
        gc.collect()
    torch.cuda.synchronize()
    assert len(model._lh_plans) == n_plans == 0
    assert torch.cuda.memory_allocated() <= mem + (1 << 20), (torch.cuda.memory_allocated(), mem)


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher must yield two ranks (the parent starts torch.distributed.run as a
    child and relays rank 0's line) or exit non-zero -- never run one rank and report it as two.  Rehearsed on this one GPU
    with the gloo transport (LH_DIST_BACKEND=gloo: both ranks share the card), a small model and two steps."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--depth", "18",
                        "--batch", "4", "--size", "128", "--no-cpu-baseline", "--no-roofline", "--no-extra"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dist_backend"] == "gloo", out
    assert out["config"]["global_batch"] == 8 and out["value"] > 0
    # the exposed cost of the gradient exchange (step with collectives - step with GradSync stubbed), printed for every N > 1
    ex = out["allreduce_exposed"]
    assert isinstance(out["allreduce_exposed_ms"], float) and ex["ms_per_step_with_collectives"] > 0 and ex["ms_per_step_collectives_stubbed"] > 0
    # a rank count the launcher did not create is refused
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--depth", "18",
                        "--batch", "4", "--size", "128", "--no-cpu-baseline", "--no-roofline", "--no-extra"],
                       env=env1, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sum_chunks_matches_a_rank_ordered_fp32_sum(dtype):
    """lh_sum_chunks (the local half of GradSync(algo="direct")): out[i] = sum_r in[r, i], fp32 accumulation in row order, one rounding;
    ragged length, eight rows = the eight ranks of a node."""
    from lighthand_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(5)
    rows, n = 8, 100003
    x = torch.randn(rows, n, generator=g).to(dtype).cuda()
    out = torch.empty(n, dtype=dtype, device="cuda")
    _lib.check(lib.lh_sum_chunks(x.data_ptr(), out.data_ptr(), rows, n, _lib.dtype_code(dtype), torch.cuda.current_stream().cuda_stream), "lh_sum_chunks")
    acc = x[0].float()
    for r in range(1, rows):
        acc = acc + x[r].float()
    assert torch.equal(out, acc.to(dtype))
    assert lib.lh_sum_chunks(x.data_ptr(), out.data_ptr(), rows, n, _lib.LH_F16, None) != 0        # fp32 / bf16 buckets only


def test_bench_gpus_2_direct_gradient_exchange():
    """The same two-rank rehearsal with --grad-algo direct (all-to-all + lh_sum_chunks + all-gather): the step runs, both ranks are seen
    and the loss after the timed steps equals the all-reduce run's (two ranks: a + b in either order is the same fp32 sum)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    losses = {}
    for algo in ("direct", "allreduce"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--depth", "18",
                            "--batch", "4", "--size", "128", "--no-cpu-baseline", "--no-roofline", "--no-extra", "--grad-algo", algo],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and algo in out["config"]["workload"], out
        losses[algo] = out["loss_after"]
    assert losses["direct"] == losses["allreduce"], losses

