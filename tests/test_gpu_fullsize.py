"""The BASELINE.json configurations at their FULL sizes (C2: R50, 256 x 256, batch 64, bf16 training; C5: R50, 384 x 384,
batch 256, fp16 inference graph), checked through properties that do not need an oracle run of that size -- exact
linearity of the backward pass in the upstream gradient, bit-stable rebuilds, exact equivariance under a permutation of
the batch, decode == oracle decode of the same heat-maps -- plus an oracle comparison on what the CPU finishes in seconds
(the whole C2 forward batch; four images of C5)."""
import numpy as np
import pytest
import torch

from conftest import resnet_cfg

pytestmark = pytest.mark.gpu


def _r50(precision, seed=9001):
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    torch.manual_seed(seed)
    return get_pose_net(resnet_cfg(50), True).cuda().set_precision(precision)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_c1_r18_256_bs8_training_step_matches_oracle():
    """BASELINE.json configs[0] at its OWN size -- SimpleBaseline-ResNet18, 256 x 256, batch 8, 21 joints, fp32, random init
    (seed 9001), synthetic images; the reference's CPU-runnable case (src/tools/train.py:13-120 with --batch_size 8) -- on the
    HIP path against the CPU oracle: train-mode forward within 1e-3 of the peak, JointsMSELoss within 1e-4 relative, arg-max
    key points EQUAL on every joint (both decoders applied to their own heat-maps), the whole-model gradient as close to an
    fp64 run of the oracle as the oracle's own fp32 run is, and after one Adam step (lr 1e-3) the weights within 2 x lr of
    the oracle's."""
    from lighthand_amd.heatmap import JointsMSELoss, get_max_preds, render_targets
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    from lighthand_amd.optim import Adam
    from oracle import models as omod
    from oracle.heatmap import get_max_preds as oracle_decode
    torch.manual_seed(9001)                                          # src/tools/train.py:15
    model = get_pose_net(resnet_cfg(18), True)
    rng = np.random.RandomState(9001)
    x = torch.from_numpy(rng.randn(8, 3, 256, 256).astype(np.float32))
    joints = torch.from_numpy(rng.uniform(20, 236, size=(8, 21, 2)).astype(np.float32))
    tgt = render_targets(joints.cuda()).cpu()
    fwd = lambda s, xx: omod.pose_resnet_forward(s, xx, 18, "pytorch", training=True)
    sd = omod.clone_state(model.state_dict())
    loss_ref, pred_ref, g32 = omod.loss_and_grads(sd, fwd, x, tgt)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    _, _, g64 = omod.loss_and_grads(sd64, fwd, x.double(), tgt.double())
    model = model.cuda().train()
    opt = Adam(model.parameters(), lr=1e-3).bind_arena(model.arena())
    pred = model(x.cuda())
    loss = JointsMSELoss(False)(pred, tgt.cuda(), None)
    loss.backward()
    got, want = pred.detach().cpu().numpy(), pred_ref.numpy()
    assert got.shape == want.shape == (8, 21, 64, 64)
    assert rel(got, want) < 1e-3, rel(got, want)
    assert abs(float(loss.detach()) - loss_ref) < 1e-4 * abs(loss_ref)
    assert np.array_equal(get_max_preds(got)[0], oracle_decode(want)[0])
    num_h = num_c = den = 0.0
    for k, p in model.named_parameters():
        gh, gc, gt = p.grad.cpu().double().numpy(), g32[k].double().numpy(), g64[k].numpy()
        num_h += ((gh - gt) ** 2).sum(); num_c += ((gc - gt) ** 2).sum(); den += (gt ** 2).sum()
    l2_h, l2_c = (num_h / den) ** 0.5, (num_c / den) ** 0.5
    print(f"C1 (R18, 256 x 256, bs 8, fp32): forward rel {rel(got, want):.2e}, gradient rel-L2 vs fp64 oracle HIP {l2_h:.2e} / CPU fp32 {l2_c:.2e}")
    assert l2_h <= 3 * l2_c + 1e-5
    opt.step()
    adam = omod.AdamState(lr=1e-3)
    adam.step(sd, g32)
    torch.cuda.synchronize()
    worst = max(float((p.detach().cpu() - sd[k]).abs().max()) for k, p in model.named_parameters())
    assert worst <= 2e-3 + 1e-6, worst                               # Adam's first step moves an element by lr x sign(g)


def test_c2_training_step_full_size_properties():
    """R50, 256 x 256, batch 64 (BASELINE.json configs[1]).  bf16, the benchmark's precision: the backward pass is EXACTLY
    linear in the heat-map gradient (2 * dheat -> 2 * every weight / BatchNorm gradient, bit for bit: every kernel of the
    backward pass at full size, split-K folds and fp64 statistics included) and a second model built from the same seed
    reproduces heat-maps and gradients bit for bit.  fp32: the train-mode forward of the whole batch within 1e-3 of the
    CPU oracle (a random-init network in train mode amplifies bf16 rounding to O(1) differences -- DESIGN.md section 4 --
    so values are compared in fp32)."""
    from oracle import models as omod
    rng = np.random.RandomState(64)
    x = torch.from_numpy(rng.randn(64, 3, 256, 256).astype(np.float32))
    dheat = torch.from_numpy((rng.randn(64, 21, 64, 64) / 64).astype(np.float32)).to(torch.bfloat16).float().cuda()
    runs = []
    for _ in range(2):
        m = _r50("bf16").train()
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        plan = m.plan(64, 256, 256, training=True, backward=True)
        hm = plan.forward(x.cuda()).clone()
        plan.backward(dheat)
        torch.cuda.synchronize()
        g1 = m.arena().flat_grad.clone()
        plan.backward(2 * dheat)
        torch.cuda.synchronize()
        g2 = m.arena().flat_grad.clone()
        assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
        assert torch.equal(g2, 2 * g1), "the backward pass must be exactly linear in the upstream gradient"
        runs.append((hm, g1))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), "rebuild from the same seed must be bit-stable"
    assert runs[0][0].shape == (64, 21, 64, 64)
    del plan, m
    m32 = _r50("fp32").train()
    with torch.no_grad():
        got = m32(x.cuda()).cpu().numpy()
        want = omod.pose_resnet_forward(sd, x, 50, training=True).numpy()
    assert rel(got, want) < 1e-3, rel(got, want)


def test_c5_inference_graph_full_size_properties():
    """R50 inference, 384 x 384, batch 256, fp16, hipGraph replay (BASELINE.json configs[4]): permuting the batch permutes
    heat-maps and keypoints EXACTLY (eval-mode BatchNorm: images are independent; tiles straddle image borders, the result
    must not care); the device decode equals the oracle's decode of the same heat-maps bit for bit; four images against
    the CPU oracle's eval-mode forward within the fp16 tolerance."""
    from lighthand_amd.runtime import InferStep
    from oracle import models as omod
    from oracle.heatmap import get_max_preds
    m = _r50("fp16").eval()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    rng = np.random.RandomState(256)
    x = torch.from_numpy(rng.randn(256, 3, 384, 384).astype(np.float32))
    step = InferStep(m, 256, 384, 384)
    preds = step(x.cuda()).clone()
    torch.cuda.synchronize()
    hm = step.heatmaps.clone()
    assert hm.shape == (256, 21, 96, 96) and step.graph is not None
    assert np.array_equal(preds.cpu().numpy(), get_max_preds(hm.cpu().numpy())[0] * 4)
    perm = torch.from_numpy(rng.permutation(256))
    preds_p = step(x[perm].cuda()).clone()
    torch.cuda.synchronize()
    assert torch.equal(step.heatmaps, hm[perm.cuda()]) and torch.equal(preds_p, preds[perm.cuda()])
    with torch.no_grad():
        want = omod.pose_resnet_forward(sd, x[:4], 50, training=False).numpy()
    err = np.abs(hm[:4].cpu().numpy() - want).max() / np.abs(want).max()
    assert err < 2e-2, err


def test_c4_hrnet_w32_full_size_properties():
    """HRNet-W32, 256 x 256, batch 32 per GPU, bf16 (BASELINE.json configs[3], one rank's share): its branches run on
    concurrent stream lanes and its weight gradients on side streams, so a race would show as run-to-run noise -- heat-maps
    and gradients of two models built from the same seed are bit-equal, and the backward pass is exactly linear in the
    heat-map gradient."""
    from lighthand_amd.modeling.hrnet.pose_hrnet import get_hrnet, hrnet_cfg
    rng = np.random.RandomState(32)
    x = torch.from_numpy(rng.randn(32, 3, 256, 256).astype(np.float32)).cuda()
    dheat = torch.from_numpy((rng.randn(32, 21, 64, 64) / 64).astype(np.float32)).to(torch.bfloat16).float().cuda()
    runs = []
    for _ in range(2):
        torch.manual_seed(9001)
        m = get_hrnet(hrnet_cfg(32), True).cuda().set_precision("bf16").train()
        plan = m.plan(32, 256, 256, training=True, backward=True)
        assert plan.n_lanes > 1
        hm = plan.forward(x).clone()
        plan.backward(dheat)
        torch.cuda.synchronize()
        g1 = m.arena().flat_grad.clone()
        plan.backward(2 * dheat)
        torch.cuda.synchronize()
        assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
        assert torch.equal(m.arena().flat_grad, 2 * g1)
        runs.append((hm, g1))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
