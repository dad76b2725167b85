"""Whole-model parity on the GPU through the drop-in API (get_pose_net / get_hrnet,
JointsMSELoss, Adam): HIP fp32 path vs the golden vectors generated from the reference
(G5 forward, G6 three-step Adam trajectory) and vs the CPU oracle on fresh inputs."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import resnet_cfg

pytestmark = pytest.mark.gpu

FP32_REL = 1e-3          # north star: within 1e-3 relative in fp32


def _build(tag):
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    from lighthand_amd.modeling.hrnet.pose_hrnet import get_hrnet, hrnet_cfg
    from oracle import models as omod
    if tag.startswith("hrnet"):
        return get_hrnet(hrnet_cfg(int(tag.split("w")[1])), True), lambda sd, x, tr: omod.hrnet_forward(sd, x, training=tr)
    if tag.startswith("mini"):        # one unit per stage: every op kind of R18 / R50, shallow enough to be well conditioned
        from lighthand_amd.modeling.simplebaseline import pose_resnet as pr
        kind = "bottleneck" if tag == "mini_bottleneck" else "basic"
        pr.resnet_spec[-1] = (kind, [1, 1, 1, 1])
        return (get_pose_net(resnet_cfg(-1), True),
                lambda sd, x, tr: omod.pose_resnet_forward(sd, x, training=tr, spec=(kind, [1, 1, 1, 1])))
    depth = {"r18": 18, "r34": 34, "r50": 50, "r50caffe": 50, "r101": 101, "r152": 152}[tag]
    style = "caffe" if tag.endswith("caffe") else "pytorch"
    return (get_pose_net(resnet_cfg(depth, style), True),
            lambda sd, x, tr: omod.pose_resnet_forward(sd, x, depth, style, training=tr))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize("tag", ["r18", "r50", "r50caffe", "hrnet_w32", "hrnet_w48"])
def test_g5_forward_fp32_matches_reference(golden_dir, tag):
    meta = json.load(open(os.path.join(golden_dir, "g5_models.json")))[tag]
    g = np.load(os.path.join(golden_dir, "g5_models.npz"))
    torch.manual_seed(meta["seed"])
    model, _ = _build(tag)
    model = model.cuda()
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    model.train()
    with torch.no_grad():
        y_tr = model(x).cpu().numpy()
    model.eval()
    with torch.no_grad():
        y_ev = model(x).cpu().numpy()
    assert y_tr.shape == g[f"{tag}_train"].shape
    assert rel(y_tr, g[f"{tag}_train"]) < FP32_REL
    assert rel(y_ev, g[f"{tag}_eval"]) < FP32_REL
    sd = model.state_dict()
    for k, v in meta["bn_after_train_fwd"].items():
        assert abs(float(sd[k].double().sum()) - v) <= 1e-4 * max(1.0, abs(v)), k
    assert int(sd["bn1.num_batches_tracked"]) == 1
    # arg-max keypoints of the HIP heatmaps equal those of the reference heatmaps
    from lighthand_amd.heatmap import get_max_preds
    from oracle.heatmap import get_max_preds as oracle_decode
    assert np.array_equal(get_max_preds(y_tr)[0], oracle_decode(y_tr)[0])


@pytest.mark.parametrize("tag", ["mini_basic", "mini_bottleneck", "r18", "r50", "hrnet_w32", "mini_basic@3x96x160", "mini_bottleneck@5x160x96"])
def test_gradients_match_oracle(tag):
    """dL/dtheta of every parameter, HIP fp32 vs autograd through the CPU oracle.

    Deep random-init BatchNorm networks are chaotic in their gradients (on the CPU oracle itself
    a 1e-7 input perturbation moves some R50 layer-4 weight gradients by 20 %), so the yardstick
    is an fp64 run of the oracle: the HIP fp32 gradients must be as close to it as the oracle's own
    fp32 run is (global rel-L2 and per-tensor median), and per tensor for the shallow 'mini' nets
    (one unit per stage: every op kind, well conditioned)."""
    from oracle import models as omod
    from lighthand_amd.heatmap import JointsMSELoss
    torch.manual_seed(11)
    b, h, w = 4, 128, 128
    if "@" in tag:                # ragged case: odd batch (pixel-tile tails), non-square map
        tag, dims = tag.split("@")
        b, h, w = (int(v) for v in dims.split("x"))
    model, fwd = _build(tag)
    rng = np.random.RandomState(5)
    x = torch.from_numpy(rng.randn(b, 3, h, w).astype(np.float32))
    tgt = torch.from_numpy(rng.rand(b, 21, h // 4, w // 4).astype(np.float32))
    sd = omod.clone_state(model.state_dict())
    loss_ref, pred_ref, g32 = omod.loss_and_grads(sd, lambda s, xx: fwd(s, xx, True), x, tgt)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    _, _, g64 = omod.loss_and_grads(sd64, lambda s, xx: fwd(s, xx, True), x.double(), tgt.double())
    model = model.cuda().train()
    pred = model(x.cuda())
    loss = JointsMSELoss(False)(pred, tgt.cuda(), None)
    loss.backward()
    assert abs(float(loss.detach()) - loss_ref) < 1e-4 * abs(loss_ref)
    assert rel(pred.detach().cpu().numpy(), pred_ref.numpy()) < FP32_REL
    num_h = num_c = den = 0.0
    eh, ec = [], []
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        gh, gc, gt = p.grad.cpu().double().numpy(), g32[k].double().numpy(), g64[k].numpy()
        eh.append(rel(gh, gt)); ec.append(rel(gc, gt))
        num_h += ((gh - gt) ** 2).sum(); num_c += ((gc - gt) ** 2).sum(); den += (gt ** 2).sum()
    l2_h, l2_c = (num_h / den) ** 0.5, (num_c / den) ** 0.5
    print(f"{tag}: grad error vs fp64 oracle: global rel-L2 HIP fp32 {l2_h:.3e} / CPU fp32 {l2_c:.3e}; "
          f"per-tensor median {np.median(eh):.3e} / {np.median(ec):.3e}, max {max(eh):.3e} / {max(ec):.3e}")
    assert l2_h <= 3 * l2_c + 1e-5
    assert np.median(eh) <= 3 * np.median(ec) + 1e-5
    if tag.startswith("mini"):
        # per tensor: a ReLU unit whose pre-activation is within fp32 rounding of 0 flips its mask; with BN in
        # front that moves d(beta) of its channel and the weight gradient of the conv before it (x_hat ~ 0 there,
        # so d(gamma) does not move) -- on either side vs fp64.  Allow such outliers on at most 10 % of tensors.
        bad = [(e1, e2) for e1, e2 in zip(eh, ec) if e1 > max(5 * e2, 1e-2)]
        assert len(bad) <= max(1, len(eh) // 10), bad


@pytest.mark.parametrize("tag", ["r18", "r50", "hrnet_w32"])
def test_g6_three_step_trajectory(golden_dir, tag):
    from lighthand_amd.heatmap import JointsMSELoss, render_targets
    from lighthand_amd.optim import Adam
    meta = json.load(open(os.path.join(golden_dir, "g6_traj.json")))[tag]
    g = np.load(os.path.join(golden_dir, "g6_traj.npz"))
    torch.manual_seed(9001)
    model, _ = _build(tag)
    model = model.cuda().train()
    opt = Adam(model.parameters(), lr=1e-3).bind_arena(model.arena())
    crit = JointsMSELoss(False)
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    tgt = render_targets(torch.from_numpy(g[f"{tag}_joints"]).cuda())[:, :, :16, :16].contiguous()
    losses = []
    for _ in range(3):
        pred = model(x)
        loss = crit(pred, tgt, None)
        losses.append(float(loss.detach()))
        opt.zero_grad()
        loss.backward()
        opt.step()
    # step 1 is a pure forward of identical weights; later steps see Adam's sign-like updates of
    # noise-level gradients (see test_gradients_match_oracle), so they agree to a few percent only
    assert abs(losses[0] - meta["losses"][0]) < 1e-4 * meta["losses"][0]
    assert np.allclose(losses, meta["losses"], rtol=3e-2), (losses, meta["losses"])
    sd = model.state_dict()
    for k, v in meta["abs_sums"].items():
        got = float(sd[k].double().abs().sum())
        assert abs(got - v) <= 2e-2 * max(1e-6, abs(v)), (k, got, v)
    print(tag, "losses", losses, "reference", meta["losses"])


@pytest.mark.parametrize("tag", ["r18", "r50", "hrnet_w32"])
def test_g6wc_well_conditioned_trajectory_at_1e3(golden_dir, tag):
    """G6 on a WELL-CONDITIONED network, held to the north star's 1e-3 on ALL three steps: the reference ran its three Adam
    steps (method.py:160-183, train.py:45-48) after the gain of the last BatchNorm of every residual branch was scaled by
    0.05 (tests/golden/make_golden.py g6wc; the keys travel with the fixture).  In that regime -- a trained residual
    network's -- the train-mode BatchNorm stack no longer amplifies rounding noise, so the HIP fp32 path must reproduce
    the reference's losses of steps 1-3 and the picked weight sums to 1e-3 (plain G6, on the default init, can only hold
    steps 2-3 to percents)."""
    from lighthand_amd.heatmap import JointsMSELoss, render_targets
    from lighthand_amd.optim import Adam
    meta = json.load(open(os.path.join(golden_dir, "g6wc_traj.json")))[tag]
    g = np.load(os.path.join(golden_dir, "g6wc_traj.npz"))
    torch.manual_seed(9001)
    model, _ = _build(tag)
    sd = model.state_dict()
    assert set(meta["scaled_keys"]) <= set(sd)
    with torch.no_grad():
        for k in meta["scaled_keys"]:
            sd[k].mul_(meta["gain_scale"])
    model = model.cuda().train()
    opt = Adam(model.parameters(), lr=1e-3).bind_arena(model.arena())
    crit = JointsMSELoss(False)
    x = torch.from_numpy(g[f"{tag}_x"]).cuda()
    tgt = render_targets(torch.from_numpy(g[f"{tag}_joints"]).cuda())[:, :, :16, :16].contiguous()
    losses = []
    for _ in range(3):
        pred = model(x)
        loss = crit(pred, tgt, None)
        losses.append(float(loss.detach()))
        opt.zero_grad()
        loss.backward()
        opt.step()
    lerr = [abs(a - b) / abs(b) for a, b in zip(losses, meta["losses"])]
    sd = model.state_dict()
    serr = {k: abs(float(sd[k].double().abs().sum()) - v) / max(1e-6, abs(v)) for k, v in meta["abs_sums"].items()}
    final = model(x).detach().cpu().numpy()            # (a fourth forward: after the sums, as in the fixture's generator)
    print(tag, "losses", losses, "reference", meta["losses"], "rel", lerr, "worst weight abs-sum rel", max(serr.values()),
          "final heat-maps rel", rel(final, g[f"{tag}_final_pred"]))
    # The fixture also carries the reference's distance FROM ITSELF (the same three steps with one CPU thread instead of eight:
    # other summation orders inside its own kernels).  R18 / R50: 1e-7 -- the 1e-3 bar stands as it is (measured here: losses
    # 3e-6 / 2e-5, weight sums 1e-5).  HRNet-W32: the reference moves 4.6e-5 (step 2) and 2.5e-3 (step 3) against itself and
    # 7.9e-2 in the final heat-maps -- its lr = 1e-3 trajectory is unstable even on these weights (loss 2.75 -> 3.71 -> 2.06) --
    # so no second implementation can be held to 1e-3 there: the bar is max(1e-3, 4 x the reference's self-distance) per step
    # (measured here: 2.6e-7, 1.2e-4, 8.1e-3; weight sums 6.6e-4 < 1e-3).
    self_l = [abs(a - b) / abs(b) for a, b in zip(meta["losses_one_thread"], meta["losses"])]
    bars = [max(1e-3, 4 * d) for d in self_l]
    assert lerr[0] < 1e-4 and all(e < b for e, b in zip(lerr, bars)), (lerr, bars)
    assert max(serr.values()) < 1e-3, serr
    # after the third update (not part of the 1e-3 contract: Adam's third step moves an element whose gradient is rounding noise
    # by +-lr): R18 6.8e-3, R50 3.3e-4, HRNet 8.2e-2 with a reference self-distance of 7.9e-2
    assert rel(final, g[f"{tag}_final_pred"]) < max(2e-2, 2 * meta["final_pred_self_rel"])


def test_torch_adam_also_drives_the_model():
    """The reference loop uses torch.optim.Adam (train.py:45-48): it must work unchanged."""
    from lighthand_amd.heatmap import JointsMSELoss
    torch.manual_seed(3)
    model, _ = _build("r18")
    model = model.cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = torch.randn(2, 3, 64, 64, device="cuda")
    tgt = torch.rand(2, 21, 16, 16, device="cuda")
    l0 = None
    for _ in range(4):
        loss = JointsMSELoss(False)(model(x), tgt, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
        l0 = l0 or float(loss.detach())
    assert float(loss.detach()) < l0


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_reduced_precision_trains_like_fp32(precision):
    """bf16 / fp16 activations+weights (fp32 accumulation, fp32 BN statistics, fp32 master weights)
    are new relative to the fp32-only reference.  Outputs of a random-init train-mode
    network are chaotic (above), so fidelity is declared on behaviour: over 25 Adam steps on one
    fixed batch the loss curve follows the fp32 HIP run (final loss within 25 %)."""
    from lighthand_amd.heatmap import JointsMSELoss, render_targets
    from lighthand_amd.optim import Adam
    curves = {}
    for prec in ("fp32", precision):
        torch.manual_seed(0)
        model, _ = _build("r50")
        model = model.cuda().train().set_precision(prec)
        opt = Adam(model.parameters(), lr=1e-3).bind_arena(model.arena())
        rng = np.random.RandomState(1)
        x = torch.from_numpy(rng.randn(8, 3, 128, 128).astype(np.float32)).cuda()
        joints = torch.from_numpy(rng.uniform(10, 118, size=(8, 21, 2)).astype(np.float32)).cuda()
        tgt = render_targets(joints)[:, :, :32, :32].contiguous()
        crit, losses = JointsMSELoss(False), []
        for _ in range(25):
            loss = crit(model(x), tgt, None)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        curves[prec] = losses
    a, b = curves["fp32"], curves[precision]
    print(precision, "fp32 first/last", a[0], a[-1], "|", precision, "first/last", b[0], b[-1])
    assert abs(b[0] - a[0]) < 0.05 * a[0]
    assert a[-1] < 0.5 * a[0] and b[-1] < 0.5 * b[0]
    assert abs(b[-1] - a[-1]) < 0.25 * a[-1]


@pytest.mark.parametrize("tag,shape", [("r18", (3, 3, 96, 160)), ("r34", (1, 3, 64, 64)), ("r50", (5, 3, 128, 96)),
                                       ("r50caffe", (2, 3, 192, 64)), ("r101", (2, 3, 64, 96)), ("r152", (3, 3, 128, 96)),
                                       ("hrnet_w32", (3, 3, 64, 96)), ("hrnet_w48", (1, 3, 128, 128))])
def test_forward_odd_shapes_match_oracle(tag, shape):
    """Ragged cases the fixed goldens do not hold: odd batch sizes (pixel-tile tails), batch 1, non-square and
    non-power-of-two maps, every block kind -- train- and eval-mode forward in fp32 vs the CPU oracle on the same
    seeded weights, plus bf16 eval within the 16-bit tolerance."""
    torch.manual_seed(11)
    model, oracle_fwd = _build(tag)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.from_numpy(np.random.RandomState(4).randn(*shape).astype(np.float32))
    with torch.no_grad():
        want_tr = oracle_fwd({k: v.clone() for k, v in sd.items()}, x, True).numpy()
        want_ev = oracle_fwd({k: v.clone() for k, v in sd.items()}, x, False).numpy()
    model = model.cuda()
    with torch.no_grad():
        model.train()
        got_tr = model(x.cuda()).cpu().numpy()
        model.load_state_dict(sd)                      # undo the running-statistics update of the train-mode pass
        model.eval()
        got_ev = model(x.cuda()).cpu().numpy()
        got_bf = model.set_precision("bf16")(x.cuda()).cpu().numpy()
    assert got_tr.shape == want_tr.shape == (shape[0], 21, shape[2] // 4, shape[3] // 4)
    if tag in ("r101", "r152"):
        # 100+ train-mode BN layers at random init amplify fp32 rounding itself: the CPU fp32 oracle sits 1e-3 .. 3e-3 from
        # an fp64 evaluation of the same network.  Yardstick: as close to fp64 as the CPU fp32 oracle is (x3), eval at 1e-3.
        with torch.no_grad():
            sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
            want64 = oracle_fwd(sd64, x.double(), True).numpy()
        assert rel(got_tr, want64) < max(FP32_REL, 3 * rel(want_tr, want64)), (rel(got_tr, want64), rel(want_tr, want64))
        assert rel(got_ev, want_ev) < FP32_REL
    else:
        assert rel(got_tr, want_tr) < FP32_REL and rel(got_ev, want_ev) < FP32_REL
    assert rel(got_bf, want_ev) < 5e-2


def _trained_like(sd, g=0.05, seed=3):
    """Weights in the regime of a TRAINED residual network: BN gains random in [0.5, 1.5] except the last BN of every
    residual branch (x g: the branch is a small correction of its shortcut), BN offsets ~ 0.2 N(0,1).  Gradients of such a
    network are well conditioned (CPU fp32 vs fp64: global rel-L2 2.6e-4 for R50, 4.6e-4 for HRNet-W32), unlike the
    default random init, whose train-mode BN stack amplifies rounding noise to percents."""
    rng = np.random.RandomState(seed)
    out = {}
    for k, v in sd.items():
        v = v.detach().clone()
        if v.dim() == 1 and k.endswith(".weight"):
            last = k.endswith("bn3.weight") or (".bn2.weight" in k and (k[:-len("bn2.weight")] + "bn3.weight") not in sd)
            v = torch.from_numpy(((g if last else 1.0) * (0.5 + rng.rand(v.numel()))).astype(np.float32))
        elif v.dim() == 1 and k.endswith(".bias") and k.replace(".bias", ".running_mean") in sd:
            v = torch.from_numpy((0.2 * rng.randn(v.numel())).astype(np.float32))
        out[k] = v
    return out


@pytest.mark.parametrize("tag", ["r50", "r50caffe", "hrnet_w32"])
def test_gradients_well_conditioned_case_at_1e3(tag):
    """Whole-model dL/dtheta at the north-star tolerance: on trained-like weights (see _trained_like) the HIP fp32
    gradients agree with the fp64 oracle to 1e-3 in global relative L2 (every parameter tensor of the model in one
    vector), and per tensor they are as accurate as the CPU fp32 oracle (median within 2x)."""
    from oracle import models as omod
    from lighthand_amd.heatmap import JointsMSELoss
    torch.manual_seed(11)
    model, fwd = _build(tag)
    model.load_state_dict(_trained_like(model.state_dict()))
    rng = np.random.RandomState(5)
    b, h, w = 4, 128, 128
    x = torch.from_numpy(rng.randn(b, 3, h, w).astype(np.float32))
    tgt = torch.from_numpy(rng.rand(b, 21, h // 4, w // 4).astype(np.float32))
    sd = omod.clone_state(model.state_dict())
    loss_ref, pred_ref, g32 = omod.loss_and_grads(sd, lambda s, xx: fwd(s, xx, True), x, tgt)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    _, _, g64 = omod.loss_and_grads(sd64, lambda s, xx: fwd(s, xx, True), x.double(), tgt.double())
    model = model.cuda().train()
    pred = model(x.cuda())
    loss = JointsMSELoss(False)(pred, tgt.cuda(), None)
    loss.backward()
    assert abs(float(loss.detach()) - loss_ref) < 1e-4 * abs(loss_ref)
    assert rel(pred.detach().cpu().numpy(), pred_ref.numpy()) < FP32_REL
    num_h = num_c = den = 0.0
    eh, ec = [], []
    for k, p in model.named_parameters():
        gh, gc, gt = p.grad.cpu().double().numpy(), g32[k].double().numpy(), g64[k].numpy()
        eh.append(rel(gh, gt)); ec.append(rel(gc, gt))
        num_h += ((gh - gt) ** 2).sum(); num_c += ((gc - gt) ** 2).sum(); den += (gt ** 2).sum()
    l2_h, l2_c = (num_h / den) ** 0.5, (num_c / den) ** 0.5
    print(f"{tag} (trained-like weights): grad error vs fp64 oracle: global rel-L2 HIP fp32 {l2_h:.3e} / CPU fp32 {l2_c:.3e}; "
          f"per-tensor median {np.median(eh):.3e} / {np.median(ec):.3e}, max {max(eh):.3e} / {max(ec):.3e}")
    assert l2_h < FP32_REL
    assert np.median(eh) <= 2 * np.median(ec) + 1e-5


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(3, 72, 104), (2, 256, 256)])
def test_direct_training_stem_matches_the_tiled_kernel(precision, shape, monkeypatch):
    """lh_stem_conv -- conv1 of a training plan on the direct kernel (weights in registers, input patch in LDS; pose_resnet.py:
    151-152) -- writes the convolution output of the tiled kernel BIT FOR BIT (ragged tiles included) and statistics rows
    whose totals are its column sums; the first training-mode heat-maps agree to the rounding the statistics' summation
    order allows."""
    from lighthand_amd import _lib
    lib = _lib.load()
    b, h, w = shape
    rng = np.random.RandomState(8)
    x = torch.from_numpy(rng.randn(b, 3, h, w).astype(np.float32)).cuda()
    got = {}
    for direct in ("0", "1"):
        monkeypatch.setenv("LH_STEM_DIRECT", direct)
        monkeypatch.setenv("LH_AUTOTUNE", "0")
        torch.manual_seed(11)
        model, _ = _build("r18")
        model = model.cuda().set_precision(precision).train()
        plan = model.plan(b, h, w, training=True, backward=True)            # (a plan with a backward pass keeps every activation)
        n_direct = sum(1 for c in plan.fwd if getattr(c, "fn", None) is lib.lh_stem_conv)
        assert n_direct == int(direct)
        out = plan.forward(x).detach().clone()
        torch.cuda.synchronize()
        stem = [nd for kind, nd in plan.nodes if kind == "conv"][0]["y"]                 # conv1: the first convolution of the graph
        rows = stem.stats_rows
        st = stem.stats[:rows * 2 * 64].view(rows, 2, 64).double().sum(0)
        got[direct] = (stem.buf.clone(), st.clone(), out)
    assert torch.equal(got["0"][0], got["1"][0])
    conv = got["1"][0].double()
    want = torch.stack([conv.reshape(-1, 64).sum(0), (conv * conv).reshape(-1, 64).sum(0)])
    assert torch.allclose(got["1"][1], want, rtol=1e-5, atol=1e-3), float((got["1"][1] - want).abs().max())
    assert torch.allclose(got["0"][1], got["1"][1], rtol=1e-5, atol=1e-3)
    assert rel(got["0"][2].cpu().numpy(), got["1"][2].cpu().numpy()) < 2e-2


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_bn_backward_gate_in_the_data_gradient_matches_the_separate_reduce_pass(precision, monkeypatch):
    """lh_igemm_gated: the data gradient that writes the gradient of a = relu(BN(x)) stores the ReLU-gated gradient and the
    BatchNorm-backward partial sums of its tiles; the node's backward then runs without its reduce pass (loss.backward()
    through `out = self.relu(self.bn2(self.conv2(out)))`, pose_resnet.py:60-66).  Same model, same batch, gate on / off:
    the launches change (gated data gradients appear, as many reduce passes disappear), the whole-model gradient agrees to
    the rounding of the 16-bit activations gradients (the fp32 partial sums are formed in another order: a few gradients
    round to the neighbouring 16-bit value)."""
    from lighthand_amd import _lib
    from lighthand_amd.heatmap import JointsMSELoss
    lib = _lib.load()
    rng = np.random.RandomState(5)
    x = torch.from_numpy(rng.randn(4, 3, 128, 128).astype(np.float32)).cuda()
    tgt = torch.from_numpy(rng.rand(4, 21, 32, 32).astype(np.float32)).cuda()
    grads, gated = {}, {}
    for gate in ("0", "1"):
        monkeypatch.setenv("LH_BN_GATE", gate)
        monkeypatch.setenv("LH_AUTOTUNE", "0")
        torch.manual_seed(11)
        model, _ = _build("r50")
        model.load_state_dict(_trained_like(model.state_dict()))
        model = model.cuda().set_precision(precision).train()
        pred = model(x)
        # the configuration the product SHIPS: fp16 plans always train with the static loss scale (TrainStep default 1024; a
        # power of two, so scaling the loss here and dividing the gradients is exact), bf16 plans with none
        scale = 1024.0 if precision == "fp16" else 1.0
        (JointsMSELoss(False)(pred, tgt, None) * scale).backward()
        plan = model.plan(4, 128, 128, training=True, backward=True)
        gated[gate] = sum(1 for c in plan.bwd if getattr(c, "fn", None) is lib.lh_igemm_gated)
        grads[gate] = torch.cat([p.grad.flatten().double() for p in model.parameters()]).cpu() / scale
    assert gated["0"] == 0 and gated["1"] >= 10, gated          # R50: most bn1 / bn2 nodes sit in front of a tiled data gradient
    err = float((grads["1"] - grads["0"]).norm() / grads["0"].norm())
    print(f"{precision}: {gated['1']} gated data gradients; whole-model gradient, gate on vs off: rel-L2 {err:.3e}")
    # bound per dtype = 2 x the measured value (round 5; static kernel choice, so the figure is reproducible)
    assert err < GATE_BOUND[precision], err


GATE_BOUND = {"bf16": 1.1e-3, "fp16": 1.5e-4}      # 2 x measured: bf16 5.29e-4; fp16 WITH the shipped loss scale 7.08e-5 (6e-3 without it, round 4)


def test_c2_r50_bf16_gradients_vs_fp32_oracle():
    """The timed configuration's arithmetic (R50, bf16 activations / weight packs, fp32 accumulation and fp32 master
    gradients) against the fp32 CPU oracle, whole-model dL/dtheta at batch 8, 128 x 128, on the well-conditioned
    trained-like weights of test_gradients_well_conditioned_case_at_1e3 (random-init or over-fitted train-mode BN stacks
    amplify ANY rounding to tens of percent, CPU fp32 included, and would measure conditioning, not kernels).
    Declared bf16 tolerance: global relative L2 over all parameters <= 5e-2 (measured 3.2e-2),
    i.e. cosine >= 0.998 between the two whole-model gradients; loss within 1e-3 relative.  Per TENSOR the errors are
    larger (median 0.3): on this network the fp32 path's per-tensor median is 1e-4 = 1.7e3 units of fp32 round-off, and the
    same amplification of bf16's 2^-9 is O(0.3) -- a property of train-mode BatchNorm gradients (sums with heavy
    cancellation), not of the kernels; it averages out over steps (test_reduced_precision_trains_like_fp32).  The test
    therefore also requires every tensor's gradient to point the right way: per-tensor cosine, median >= 0.9."""
    from oracle import models as omod
    from lighthand_amd.heatmap import JointsMSELoss
    torch.manual_seed(11)
    model, fwd = _build("r50")
    model.load_state_dict(_trained_like(model.state_dict()))
    rng = np.random.RandomState(5)
    b, h, w = 8, 128, 128
    x = torch.from_numpy(rng.randn(b, 3, h, w).astype(np.float32))
    tgt = torch.from_numpy(rng.rand(b, 21, h // 4, w // 4).astype(np.float32))
    sd = omod.clone_state(model.state_dict())
    loss_ref, pred_ref, g32 = omod.loss_and_grads(sd, lambda s, xx: fwd(s, xx, True), x, tgt)
    model = model.cuda().set_precision("bf16").train()
    pred = model(x.cuda())
    loss = JointsMSELoss(False)(pred, tgt.cuda(), None)
    loss.backward()
    num = den = dot = nh = 0.0
    per, cos = [], []
    for k, p in model.named_parameters():
        gh, gr = p.grad.cpu().double().numpy().ravel(), g32[k].double().numpy().ravel()
        num += ((gh - gr) ** 2).sum(); den += (gr ** 2).sum(); dot += (gh * gr).sum(); nh += (gh ** 2).sum()
        per.append(float(np.sqrt(((gh - gr) ** 2).sum() / ((gr ** 2).sum() + 1e-30))))
        cos.append(float((gh * gr).sum() / (np.sqrt((gh ** 2).sum() * (gr ** 2).sum()) + 1e-30)))
    l2, gcos = float(np.sqrt(num / den)), float(dot / np.sqrt(nh * den))
    herr = rel(pred.detach().cpu().numpy(), pred_ref.numpy())
    lrel = abs(float(loss.detach()) - loss_ref) / abs(loss_ref)
    print(f"C2 arithmetic (bf16) vs fp32 oracle on trained-like weights: gradient global rel-L2 {l2:.3e} (cosine {gcos:.5f}), per-tensor rel-L2 "
          f"median {np.median(per):.3e} max {max(per):.3e}, per-tensor cosine median {np.median(cos):.4f} min {min(cos):.4f}; "
          f"heat-map err {herr:.3e}; loss rel {lrel:.3e}")
    assert l2 < 5e-2 and gcos > 0.998
    assert np.median(cos) >= 0.9
    assert herr < 5e-2            # flat random-feature maps; the forward pin is test_c2_r50_bf16_train_forward_matches_fp32_oracle
    assert lrel < 1e-3


def test_c4_hrnet_w32_fp16_gradients_vs_fp32_oracle(monkeypatch):
    """C4's TIMED dtype (bench.py extra.hrnet_w32_train_bs32: fp16 + static loss scale 1024) against the fp32 CPU oracle:
    whole-model dL/dtheta of HRNet-W32 (pose_hrnet.py:139-185, 247-265, 401-460) at batch 4, 128 x 128, on the
    well-conditioned trained-like weights, merged multi-problem launches, static kernel choice (reproducible figures).
    Round 4 measured the bf16 form of this gradient 0.29 from the fp32 plan's -- which is why fp16 is
    the timed dtype; this test holds fp16 to its own measured distance."""
    from oracle import models as omod
    from lighthand_amd.heatmap import JointsMSELoss
    monkeypatch.setenv("LH_AUTOTUNE", "0")
    torch.manual_seed(11)
    model, fwd = _build("hrnet_w32")
    model.load_state_dict(_trained_like(model.state_dict()))
    rng = np.random.RandomState(5)
    b, h, w = 4, 128, 128
    x = torch.from_numpy(rng.randn(b, 3, h, w).astype(np.float32))
    tgt = torch.from_numpy(rng.rand(b, 21, h // 4, w // 4).astype(np.float32))
    sd = omod.clone_state(model.state_dict())
    loss_ref, pred_ref, g32 = omod.loss_and_grads(sd, lambda s, xx: fwd(s, xx, True), x, tgt)
    stats = {}
    for prec, scale in (("fp16", 1024.0), ("bf16", 1.0)):
        m = model.cuda().set_precision(prec).train()
        m.zero_grad(set_to_none=True)
        pred = m(x.cuda())
        loss = JointsMSELoss(False)(pred, tgt.cuda(), None)
        (loss * scale).backward()
        num = den = dot = nh = 0.0
        cos = []
        for k, p in m.named_parameters():
            gh, gr = p.grad.cpu().double().numpy().ravel() / scale, g32[k].double().numpy().ravel()
            num += ((gh - gr) ** 2).sum(); den += (gr ** 2).sum(); dot += (gh * gr).sum(); nh += (gh ** 2).sum()
            cos.append(float((gh * gr).sum() / (np.sqrt((gh ** 2).sum() * (gr ** 2).sum()) + 1e-30)))
        stats[prec] = (float(np.sqrt(num / den)), float(dot / np.sqrt(nh * den)), float(np.median(cos)), float(min(cos)),
                       rel(pred.detach().cpu().numpy(), pred_ref.numpy()), abs(float(loss.detach()) - loss_ref) / abs(loss_ref))
        print(f"C4 arithmetic ({prec}, loss scale {scale:g}) vs fp32 oracle, HRNet-W32 trained-like weights: gradient global rel-L2 {stats[prec][0]:.3e} "
              f"(cosine {stats[prec][1]:.5f}), per-tensor cosine median {stats[prec][2]:.4f} min {stats[prec][3]:.4f}; heat-map err {stats[prec][4]:.3e}; "
              f"loss rel {stats[prec][5]:.3e}")
    l2, gcos, cmed, cmin, herr, lrel = stats["fp16"]
    # measured (round 5, static kernel choice): fp16 global rel-L2 9.32e-2 (cosine 0.99565), per-tensor cosine median 0.9975 /
    # min 0.9899, heat-maps 3.9e-3, loss 2.7e-5; bf16 beside it: 2.55e-1 (cosine 0.9675), heat-maps 3.4e-2
    assert l2 < C4_FP16_GRAD_L2 and gcos > 0.99 and cmed >= 0.99 and cmin >= 0.97, stats
    assert herr < 1e-2 and lrel < 1e-3, stats
    assert stats["fp16"][0] < stats["bf16"][0], stats          # the reason fp16 is C4's timed dtype


C4_FP16_GRAD_L2 = 1.4e-1       # 1.5 x the measured 9.32e-2


def test_hrnet_branch_batching_is_bit_exact_and_cuts_launches(monkeypatch):
    """HRNet's parallel branches as multi-problem launches (lh_igemm_multi / lh_bn_finalize_multi / lh_fuse_fwd_multi /
    lh_fuse_bwd_multi / lh_wgrad_fused_multi; pose_hrnet.py:139-185, 247-265): the merged launch lists give BIT-IDENTICAL
    heat-maps and parameter gradients to the same launches run one by one on the same plan, the batched plan issues well
    under half the C-ABI calls of the stream-lane plan (LH_BATCH=0); with the kernel choice pinned the batched and the
    stream-lane plan agree bit for bit, and the measured choice agrees with them to the bf16 tolerance (it differs in
    tiles, i.e. in the number of BN partial-sum rows)."""
    from lighthand_amd.engine import _Call
    torch.manual_seed(3)
    model, _ = _build("hrnet_w32")
    # trained-like weights: with the default random init the train-mode BN stack amplifies ANY last-bit difference between
    # two kernel choices to tens of percents of the gradient, which would make the cross-plan bound below meaningless
    model.load_state_dict(_trained_like(model.state_dict()))
    model = model.cuda().set_precision("bf16").train()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rng = np.random.RandomState(2)
    b, h, w = 4, 128, 96
    x = torch.from_numpy(rng.randn(b, 3, h, w).astype(np.float32)).cuda()
    dheat = torch.from_numpy(rng.randn(b, 21, h // 4, w // 4).astype(np.float32)).cuda()

    def run(plan, lists):
        model.load_state_dict(sd0)
        s = torch.cuda.current_stream().cuda_stream
        plan.img_nchw.copy_(x)
        plan.dout_nchw.copy_(dheat)
        plan.refresh_packs(s)
        saved = plan.fwd, plan.bwd
        plan.fwd, plan.bwd = lists
        try:
            plan.run_forward(s)
            plan.run_backward(s)
            torch.cuda.synchronize()
        finally:
            plan.fwd, plan.bwd = saved
        return plan.out_nchw.clone(), model.arena().flat_grad.clone()

    plan = model.plan(b, h, w, training=True, backward=True)
    assert plan.batch and plan._n_groups > 0
    n_multi = sum(1 for c in plan.fwd + plan.bwd if isinstance(c, _Call) and c.fn.__name__.endswith("_multi"))
    calls_b = sum(1 for c in plan.fwd + plan.bwd if isinstance(c, _Call))
    calls_u = sum(1 for c in plan.unmerged[0] + plan.unmerged[1] if isinstance(c, _Call))
    out_m, g_m = run(plan, (plan.fwd, plan.bwd))
    out_u, g_u = run(plan, plan.unmerged)
    assert torch.equal(out_m, out_u) and torch.equal(g_m, g_u)
    assert float(g_m.abs().sum()) > 0 and torch.isfinite(g_m).all()
    import collections
    left = collections.Counter(c.fn.__name__ for c in plan.fwd + plan.bwd if isinstance(c, _Call))
    print(f"HRNet-W32 batched plan: {calls_b} C-ABI calls ({n_multi} multi-problem) instead of {calls_u}: {dict(left)}")
    assert n_multi > 100 and calls_b < 0.6 * calls_u
    # against the stream-lane plan: with the kernel choice pinned (static defaults in both plans) the two schedules run
    # the same kernels on the same tiles and agree BIT FOR BIT
    monkeypatch.setenv("LH_AUTOTUNE", "0")
    model._lh_plans.clear()
    static = model.plan(b, h, w, training=True, backward=True)
    assert static.batch
    out_s, g_s = run(static, (static.fwd, static.bwd))
    monkeypatch.setenv("LH_BATCH", "0")
    model._lh_plans.clear()
    lanes = model.plan(b, h, w, training=True, backward=True)
    assert not lanes.batch
    out_l, g_l = run(lanes, (lanes.fwd, lanes.bwd))
    assert torch.equal(out_s, out_l) and torch.equal(g_s, g_l)
    # measured vs static kernel choice: two equally valid bf16 evaluations (different tiles -> different BN partial-sum rows
    # -> last-bit differences of the statistics and of bf16 roundings downstream).  The heat-maps stay at the bf16 level;
    # the GRADIENT of this network in bf16 does not: a last-bit difference in the forward pass moves the early layers'
    # gradients by tens of percent (both plans sit 0.29 from the fp32 plan's gradient, and 0.2 from each other as soon as
    # their tiles differ; round 4, DESIGN.md section 4 -- which is why C4 is timed in fp16).  What a kernel choice must NOT
    # do is move the plan further from the fp32 gradient than the static choice is.
    assert rel(out_m.cpu().numpy(), out_l.cpu().numpy()) < 3e-2
    model.set_precision("fp32")
    model._lh_plans.clear()
    ref = model.plan(b, h, w, training=True, backward=True)
    _, g_f = run(ref, (ref.fwd, ref.bwd))
    e_m = float((g_m.double() - g_f.double()).norm() / g_f.double().norm())
    e_l = float((g_l.double() - g_f.double()).norm() / g_f.double().norm())
    d = (g_m - g_l).double()
    print("measured vs static kernel choice: heat-maps", rel(out_m.cpu().numpy(), out_l.cpu().numpy()), "gradients", float(d.norm() / g_l.double().norm()),
          f"; bf16 gradient vs the fp32 plan: measured {e_m:.3f}, static {e_l:.3f}")
    assert e_m < 1.1 * e_l + 0.01, (e_m, e_l)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 256, 256), (3, 100, 132), (1, 64, 48), (5, 72, 200)])
def test_fused_inference_stem_is_bit_identical_to_conv_then_pool(shape, precision):
    """Inference plans run maxpool(relu(bn1(conv1(x)))) (pose_resnet.py:151-156) as ONE launch (lh_stem_pool: the 64-channel
    convolution output is never written).  Same K order, same epilogue arithmetic, maximum over the rounded values: the
    heat-maps must equal those of the plan that keeps convolution and max-pool apart BIT FOR BIT -- on full tiles, on ragged
    sizes (partial 8 x 8 pooled tiles, odd convolution sizes) and with non-trivial running statistics -- and match the
    oracle like any other eval-mode forward."""
    from lighthand_amd.engine import Plan
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    from oracle import models as omod
    n, h, w = shape
    rng = np.random.RandomState(7)
    x = torch.from_numpy(rng.randn(n, 3, h, w).astype(np.float32)).cuda()
    outs, calls = [], []
    for fuse in (True, False):
        Plan.fuse_stem = fuse
        try:
            torch.manual_seed(21)
            m = get_pose_net(resnet_cfg(18), True)
            with torch.no_grad():                                      # running statistics away from (0, 1): the folded affine matters
                m.bn1.running_mean.uniform_(-0.5, 0.5)
                m.bn1.running_var.uniform_(0.5, 2.0)
                m.bn1.weight.uniform_(0.5, 1.5)
                m.bn1.bias.uniform_(-0.3, 0.3)
            sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
            m = m.cuda().set_precision(precision).eval()
            with torch.no_grad():
                outs.append(m(x).clone())
            plan = m.plan(n, h, w, training=False, backward=False)
            calls.append([getattr(c, "what", "") for c in plan.fwd])
        finally:
            Plan.fuse_stem = True
    assert any("stem fwd + maxpool" in c for c in calls[0]) and not any("maxpool fwd" in c for c in calls[0])
    assert any("maxpool fwd" in c for c in calls[1])
    assert torch.equal(outs[0], outs[1])
    with torch.no_grad():
        want = omod.pose_resnet_forward(sd, x.cpu(), 18, training=False).numpy()
    assert rel(outs[0].cpu().numpy(), want) < (3e-2 if precision == "bf16" else 4e-3)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("tag,shape", [("r50", (2, 256, 256)), ("r50", (3, 104, 136)), ("r50caffe", (1, 72, 200)), ("r50", (5, 64, 48))])
def test_fused_inference_bottleneck_is_bit_identical(tag, shape, precision):
    """Inference plans run the stride-1 bottlenecks of the first ResNet stage -- conv1 1x1 -> bn1 -> relu -> conv2 3x3 -> bn2 ->
    relu -> conv3 1x1 -> bn3, + residual, relu (pose_resnet.py:61-99) -- as ONE launch each (lh_bottleneck_infer: the 64-channel
    intermediates never leave LDS).  Same K order and epilogue arithmetic as the three launches: the heat-maps of the plan with
    the fusion must equal those of the plan without it BIT FOR BIT -- on full 16 x 16 tiles, on ragged sizes, for the block behind
    the projection shortcut (its residual is the projection's output) and the identity blocks, with non-trivial running
    statistics -- and match the oracle like any other eval-mode forward."""
    from lighthand_amd.engine import Plan
    from oracle import models as omod
    n, h, w = shape
    rng = np.random.RandomState(17)
    x = torch.from_numpy(rng.randn(n, 3, h, w).astype(np.float32)).cuda()
    outs, calls = [], []
    for fuse in (True, False):
        Plan.fuse_bottleneck = fuse
        try:
            torch.manual_seed(23)
            m, fwd = _build(tag)
            with torch.no_grad():                                      # running statistics away from (0, 1): the folded affines matter
                for k, v in m.state_dict().items():
                    if k.startswith("layer1.") and k.endswith("running_mean"):
                        v.uniform_(-0.3, 0.3)
                    elif k.startswith("layer1.") and k.endswith("running_var"):
                        v.uniform_(0.5, 2.0)
                    elif k.startswith("layer1.") and ".bn" in k and k.endswith(".bias"):
                        v.uniform_(-0.3, 0.3)
            sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
            m = m.cuda().set_precision(precision).eval()
            with torch.no_grad():
                outs.append(m(x).clone())
            plan = m.plan(n, h, w, training=False, backward=False)
            calls.append([getattr(c, "what", "") for c in plan.fwd])
        finally:
            Plan.fuse_bottleneck = True
    assert sum("bottleneck fwd" in c for c in calls[0]) == 3 and not any("bottleneck fwd" in c for c in calls[1])
    assert len(calls[1]) - len(calls[0]) == 6                          # three blocks x two launches fewer
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
    with torch.no_grad():
        want = fwd(sd, x.cpu(), False).numpy()
    assert rel(outs[0].cpu().numpy(), want) < (3e-2 if precision == "bf16" else 4e-3)


_SLICE_SNIPPET = """
import hashlib, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np, torch
from conftest import resnet_cfg
from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
torch.manual_seed(31)
m = get_pose_net(resnet_cfg(18), True).cuda().set_precision("bf16").train()
x = torch.from_numpy(np.random.RandomState(3).randn(2, 3, 64, 64).astype(np.float32)).cuda()
h = hashlib.sha256()
for _ in range(2):                                    # two forwards: the running statistics move twice
    out = m(x)
    h.update(out.detach().float().cpu().numpy().tobytes())
torch.cuda.synchronize()
for k, v in sorted(m.state_dict().items()):
    if "running" in k or "num_batches" in k:
        h.update(v.detach().cpu().numpy().tobytes())
plan = next(iter(m._lh_plans.values()))
for kind, nd in plan.nodes:                            # every BatchNorm's folded scale / shift / saved mean / invstd
    pass
print("SHA", h.hexdigest())
"""
