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

        def launch(self, flat_grad, bucket):
            s, e = bucket
            flat_grad[s:e] += self.peer[s:e]           # what all_reduce(SUM) over 2 ranks leaves behind
            self.launched.append(bucket)

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
