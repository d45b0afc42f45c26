"""GPU parity at the BASELINE.json configurations.

 * config 1 (src/unet.py UNet() defaults, 1x1x256x256, seed 0): values captured from the reference.
 * config 2 shapes (EGM-UNet(3,2,32), 8x3x512x512): the oracle is too slow for a direct comparison at this size, so the
   full-size checks are size-independent properties (bitwise run-to-run determinism, fp32-vs-bf16 agreement, conv
   linearity at the largest layer shapes) plus a direct oracle comparison on a 2-image slice in eval mode (mIoU check).
"""
import numpy as np
import pytest
import torch

from helpers import assert_close, load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"
# bf16-vs-fp32 whole-model gradient bounds at the headline shape (measured values in the test's print line)
# Measured on MI355X at random init over the 262 tensors above 1e-6 of the largest gradient norm: median 0.531, p90 0.682; worst / min cosine
# over the tensors that carry weight (>= 64 elements, norm >= 1e-4 of the largest) 0.66 / 0.78 -- and 0.002 (out_conv) / 0.004-0.012 (up4) at the
# top of the network: the error GROWS with depth from the loss (tools/bf16_grad_diag.py prints it layer by layer, also for a fixed
# dL/dlogits).  Mechanism: a forward perturbation of relative size e flips ~0.8 e of the ReLU / max-pool / arg-max decisions, and a flipped
# element is an O(1) error, so each rectifier turns e ~ 1 % (bf16 keeps 8 bits, logits differ by 2.8 %) into sqrt(0.008) ~ 9 % of white
# gradient noise; a dozen of them in a row add up in quadrature.  Inherent to 8-bit-mantissa activations (the reference trains in fp16,
# 10 bits), not to a kernel: the fp32 path of the same kernels matches the reference's gradients to 2e-5.
BF16_GRAD_MEDIAN, BF16_GRAD_P90, BF16_GRAD_WORST, BF16_GRAD_MIN_COS = 0.62, 0.78, 0.90, 0.65


def synth(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, h, w, generator=g)
    t = torch.zeros(n, h, w, dtype=torch.int64)
    for i in range(n):
        cy, cx = torch.randint(h // 4, 3 * h // 4, (2,), generator=g).tolist()
        t[i, cy - h // 8:cy + h // 8, cx - w // 6:cx + w // 6] = 1
    t[torch.rand(n, h, w, generator=g) < 0.01] = 255
    return x, t


def test_config1_unet_default_matches_reference_values():
    """BASELINE configs[0]: UNet() defaults under torch.manual_seed(0), x = randn(1,1,256,256), t = randint(0,2)."""
    from egm_unet_amd import UNet
    from egm_unet_amd.train_utils import criterion
    from egm_unet_amd.train_utils.dice_coefficient_loss import fused_criterion
    fx = load_fixture("config1_unet_default")
    torch.manual_seed(0)
    m = UNet()                                           # same init as the reference under the same seed
    x = torch.randn(1, 1, 256, 256); t = torch.randint(0, 2, (1, 256, 256))
    m.to(DEV).train()
    out = m(x.to(DEV))["out"]
    assert_close(out.detach().mean().cpu(), fx["out_mean"], rtol=1e-3, atol=1e-5, what="logit mean")
    assert_close(out.detach()[0, :, 100:116, 100:116].cpu(), fx["out_crop"], rtol=1e-3, atol=1e-4, what="logit crop")
    loss, terms = fused_criterion(out, t.to(DEV), None, dice=True, ignore_index=255, return_terms=True)
    assert_close(terms[2].cpu(), fx["dice_loss"], rtol=1e-4, atol=1e-6, what="dice loss")      # dice_loss(out, build_target(t))
    assert_close(loss.detach().cpu(), fx["criterion"], rtol=1e-4, atol=1e-5, what="criterion")
    assert_close(criterion({"out": out}, t.to(DEV), None, num_classes=2, ignore_index=255).detach().cpu(), fx["criterion"], 1e-4, 1e-5)


@pytest.fixture(scope="module")
def big_model():
    from egm_unet_amd import GRFBUNet
    torch.manual_seed(0)
    m = GRFBUNet(3, 2, base_c=32).to(DEV)
    return m


def _train_step_outputs(m, x, t, dtype):
    from egm_unet_amd.train_utils import criterion
    m.train().set_compute_dtype(dtype)
    m.zero_grad(set_to_none=True)
    lw = torch.tensor([1.0, 2.0], device=DEV)
    out = m(x)["out"]
    loss = criterion({"out": out}, t, lw, num_classes=2, ignore_index=255)
    loss.backward()
    return out.detach(), loss.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}


def test_fullsize_step_is_bitwise_deterministic_and_bf16_tracks_fp32(big_model):
    """8x3x512x512, base_c=32: two identical steps give bit-identical logits/loss/gradients (no atomics in the model path);
    bf16 logits stay close to fp32 and the argmax masks agree on >= 97 % of the pixels at random init."""
    import copy
    x, t = synth(8, 512, 512, 1)
    x, t = x.to(DEV), t.to(DEV)
    sd = copy.deepcopy(big_model.state_dict())
    o1, l1, g1 = _train_step_outputs(big_model, x, t, torch.bfloat16)
    big_model.load_state_dict(sd)
    o2, l2, g2 = _train_step_outputs(big_model, x, t, torch.bfloat16)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    bad = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    assert not bad, bad[:5]
    big_model.load_state_dict(sd)
    o3, l3, g3 = _train_step_outputs(big_model, x, t, torch.float32)
    rel = float((o1 - o3).norm() / o3.norm())
    agree = float((o1.argmax(1) == o3.argmax(1)).float().mean())
    assert rel < 0.1 and agree > 0.97, (rel, agree)
    assert abs(float(l1) - float(l3)) < 2e-2 * abs(float(l3))
    assert all(torch.isfinite(v).all() for v in g1.values())
    # whole-model GRADIENTS of the benchmarked dtype against the fp32 path (itself pinned to the reference's gradients by the
    # fixtures: median rel-L2 2e-5): per-tensor rel-L2 and cosine over all 333 parameters at the headline shape, where every
    # BatchNorm averages >= 8192 pixels (the 64 x 64 fixture's deepest levels average 32, which makes its bf16 gradients a noise test)
    gmax = max(float(v.norm()) for v in g3.values())
    rels, cosines = [], []
    for k in g3:
        a, b = g1[k].double().flatten(), g3[k].double().flatten()
        if float(b.norm()) < 1e-6 * gmax:                 # analytically ~zero (conv bias in front of a train-mode BatchNorm)
            assert float(a.norm()) < 1e-2 * gmax, (k, float(a.norm()))
            continue
        rels.append((float((a - b).norm() / b.norm()), k))
        # worst case / direction: tensors that carry weight (>= 64 elements, norm >= 1e-4 of the largest); few-element gate
        # parameters and gradients four orders below the rest are noise-dominated and enter the aggregate bars only
        if a.numel() >= 64 and float(b.norm()) >= 1e-4 * gmax:
            cosines.append((float(torch.dot(a, b) / (a.norm() * b.norm())), k))
    heavy = {k for _, k in cosines}
    worst_big = max((r, k) for r, k in rels if k in heavy)
    rels.sort(); cosines.sort()
    med, p90, worst = rels[len(rels) // 2][0], rels[int(len(rels) * 0.9)][0], worst_big
    print("bf16 vs fp32 whole-model gradients at 8x3x512x512: %d tensors, rel-L2 median %.4f, p90 %.4f, worst %.4f (%s); min cosine %.4f (%s)"
          % (len(rels), med, p90, worst[0], worst[1], cosines[0][0], cosines[0][1]))
    assert len(rels) > 150
    assert med < BF16_GRAD_MEDIAN and p90 < BF16_GRAD_P90 and worst[0] < BF16_GRAD_WORST and cosines[0][0] > BF16_GRAD_MIN_COS, (med, p90, worst, cosines[0])
    top = dict((k, r) for r, k in rels)
    assert top["out_conv.0.weight"] < 0.01 and top["up4.conv.3.weight"] < 0.03 and top["up4.conv.0.weight"] < 0.04, \
        ("the layers next to the loss must be accurate: the error has to come from depth", top["out_conv.0.weight"], top["up4.conv.3.weight"])
    big_model.load_state_dict(sd)


def test_fullsize_conv_is_linear():
    """conv(a + b) == conv(a) + conv(b) at the largest layer shape (up4.conv.0: 64 -> 32 at 8x512x512), fp32 path."""
    from egm_unet_amd import ops
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(32, 64, 3, 3, generator=g) / 24).to(DEV)
    a = torch.randn(8, 512, 512, 64, generator=g).to(DEV)
    b = torch.randn(8, 512, 512, 64, generator=g).to(DEV)
    ya, yb, yab = ops.conv2d(a, w), ops.conv2d(b, w), ops.conv2d(a + b, w)
    err = float((yab - (ya + yb)).abs().max())
    assert err < 1e-4 * float(yab.abs().max()) + 1e-5, err
    # and the bf16 MFMA path against the exact-fp32 MFMA path on bf16-representable operands
    a16 = a.bfloat16()
    y16 = ops.conv2d(a16, w.bfloat16().float())
    y32 = ops.conv2d(a16.float(), w.bfloat16().float())
    rel = float((y16.float() - y32).norm() / y32.norm())
    assert rel < 5e-3, rel


def test_eval_miou_matches_oracle_and_bf16_within_tenth_of_a_point(big_model):
    """SURVEY 8(d) mIoU check, primary form: the held-out synthetic set of 64 seeded config-2 style images at 512 x 512, eval mode, the
    same seeded weights in the oracle (CPU fp32) and in the build: fp32 logits within 1e-3 relative, argmax masks identical,
    ConfusionMatrix identical (hence mIoU identical); bf16 mIoU within +-0.1 points of it."""
    from oracle import egm_ref as R, loss_ref as L
    from egm_unet_amd.train_utils.distributed_utils import ConfusionMatrix
    sd = {k: v.detach().cpu().clone() for k, v in big_model.state_dict().items()}
    big_model.eval()
    cm, cm16 = ConfusionMatrix(2), ConfusionMatrix(2)
    ref_cm = torch.zeros(2, 2, dtype=torch.int64)
    for b in range(8):                                   # 8 batches of 8 images = the 64-image held-out set
        x, t = synth(8, 512, 512, 7700 + b)
        with torch.no_grad():
            ref = R.egm_unet_forward(sd, x, train=False)["out"]
            big_model.set_compute_dtype(torch.float32)
            out = big_model(x.to(DEV))["out"]
            big_model.set_compute_dtype(torch.bfloat16)
            o16 = big_model(x.to(DEV))["out"]
        assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-4, what="fp32 eval logits, batch %d" % b)
        assert torch.equal(out.argmax(1).cpu(), ref.argmax(1)), "argmax masks must be bit-exact on the fp32 path"
        cm.update_from_logits(t.to(DEV), out)
        cm16.update_from_logits(t.to(DEV), o16)
        ref_cm += L.confusion_matrix(t.flatten(), ref.argmax(1).flatten(), 2)
    assert np.array_equal(cm.mat.cpu().numpy(), ref_cm.numpy())
    miou = float(cm.compute()[2].mean()) * 100
    miou16 = float(cm16.compute()[2].mean()) * 100
    print("held-out 64 x 512^2: mIoU fp32 %.4f (identical confusion matrix to the oracle), bf16 %.4f" % (miou, miou16))
    assert abs(miou16 - miou) <= 0.1, (miou, miou16)
    big_model.set_compute_dtype(torch.float32)
    big_model.train()


def test_evaluate_entry_point():
    """train_utils.evaluate(model, loader, device, num_classes) -> (ConfusionMatrix, dice) like the reference's."""
    from egm_unet_amd import UNet
    from egm_unet_amd.train_utils import evaluate
    torch.manual_seed(1)
    m = UNet(3, 2, base_c=8).to(DEV)
    loader = [synth(2, 64, 64, s) for s in range(3)]
    confmat, dice = evaluate(m, loader, device=DEV, num_classes=2)
    assert confmat.mat.shape == (2, 2) and int(confmat.mat.sum()) == int(sum((t != 255).sum() for _, t in loader))
    assert 0.0 <= dice <= 1.0
    assert "mean IoU" in str(confmat)


def test_fullsize_train_step_fp32_vs_oracle(big_model):
    """One 512x512 image through EGM-UNet(3,2,32) in train mode: HIP fp32 path vs the CPU oracle.
    Logits/loss are compared with the fp32 oracle (the reference's own arithmetic).  Gradients of this badly conditioned
    problem (batch-1 BatchNorm, |loss| ~ 1e2) carry ~5e-3 fp32 rounding noise in the reference itself, so they are judged
    against the oracle run in float64: the HIP fp32 path must be as close to that truth as the fp32 CPU path is."""
    import copy
    from oracle import egm_ref as R, loss_ref as L
    x, t = synth(1, 512, 512, 5)
    sd0 = copy.deepcopy(big_model.state_dict())
    lw = torch.tensor([1.0, 2.0])
    res = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        work = {k: (v.detach().cpu().clone().to(dt) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in sd0.items()}
        for k, v in work.items():
            if v.is_floating_point() and "running_" not in k:
                v.requires_grad_(True)
        out_r = R.egm_unet_forward(work, x.to(dt), True)["out"]
        loss_r = L.criterion({"out": out_r}, t, lw.to(dt), num_classes=2, ignore_index=255)
        loss_r.backward()
        res[name] = (out_r.detach(), float(loss_r.detach()), {k: v.grad.double() for k, v in work.items() if v.is_floating_point() and v.grad is not None})
    ref_out, ref_loss = res["f32"][0], res["f32"][1]
    out, loss, grads = _train_step_outputs(big_model, x.to(DEV), t.to(DEV), torch.float32)
    assert_close(out.cpu(), ref_out, rtol=1e-3, atol=2e-4, what="fp32 train logits")
    # masks: identical except where the reference's own class margin is below the fp32 agreement level (numerical ties)
    mism = out.argmax(1).cpu() != ref_out.argmax(1)
    margin = (ref_out[:, 0] - ref_out[:, 1]).abs()
    assert int(mism.sum()) <= 1e-4 * mism.numel() and (not mism.any() or float(margin[mism].max()) < 2e-4), \
        (int(mism.sum()), float(margin[mism].max()) if mism.any() else 0.0)
    assert abs(float(loss) - ref_loss) <= 1e-5 * abs(ref_loss)
    hip_err, cpu_err = [], []
    for k, r in res["f64"][2].items():
        n = float(r.norm())
        if n > 1e-6:
            hip_err.append(float((grads[k].cpu().double() - r).norm()) / n)
            cpu_err.append(float((res["f32"][2][k] - r).norm()) / n)
    hip_err.sort(); cpu_err.sort()
    med_h, med_c = hip_err[len(hip_err) // 2], cpu_err[len(cpu_err) // 2]
    print(f"gradient rel-L2 error vs float64 oracle: HIP fp32 median {med_h:.2e} max {hip_err[-1]:.2e}; CPU fp32 median {med_c:.2e} max {cpu_err[-1]:.2e}")
    assert med_h <= 2.0 * med_c + 1e-4 and hip_err[-1] <= 2.0 * cpu_err[-1] + 1e-2, (med_h, med_c, hip_err[-1], cpu_err[-1])
    big_model.load_state_dict(sd0)


@pytest.mark.parametrize("shape", [(3, 40, 56), (2, 72, 104), (1, 90, 150)], ids=lambda s: "x".join(map(str, s)))
def test_odd_sizes_train_step_fp32_vs_oracle(shape):
    """Sizes that are not multiples of the tile / pooling factors (ragged conv tiles, odd MCA strips, the zero-pad branch of
    Up at 90x150): EGM-UNet(3,2,base_c=8), train mode, fp32, logits / loss / gradients against the CPU oracle."""
    from oracle import egm_ref as R, loss_ref as L
    from egm_unet_amd import GRFBUNet
    from egm_unet_amd.train_utils import criterion
    n, h, w = shape
    torch.manual_seed(h * 1000 + w)
    m = GRFBUNet(3, 2, base_c=8)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x, t = synth(n, h, w, 17)
    lw = torch.tensor([1.0, 2.0])
    work = {k: v.clone() for k, v in sd0.items()}
    for k, v in work.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    out_r = R.egm_unet_forward(work, x, True)["out"]
    loss_r = L.criterion({"out": out_r}, t, lw, num_classes=2, ignore_index=255)
    loss_r.backward()
    m.to(DEV).train()
    out = m(x.to(DEV))["out"]
    loss = criterion({"out": out}, t.to(DEV), lw.to(DEV), num_classes=2, ignore_index=255)
    loss.backward()
    assert out.shape == out_r.shape
    assert_close(out.detach().cpu(), out_r.detach(), rtol=2e-3, atol=5e-4, what=f"logits {shape}")
    assert abs(float(loss) - float(loss_r)) <= 2e-4 * abs(float(loss_r)) + 1e-5
    rels = []
    for k, p in m.named_parameters():
        g = work[k].grad
        if g is not None and float(g.norm()) > 1e-5:
            rels.append(float((p.grad.cpu().double() - g.double()).norm() / g.double().norm()))
    rels.sort()
    assert rels[len(rels) // 2] < 5e-3 and rels[-1] < 0.1, (rels[len(rels) // 2], rels[-1])
    for k, v in m.state_dict().items():                       # BatchNorm running statistics took the same step
        if "running_" in k:
            assert_close(v.cpu(), work[k], rtol=2e-3, atol=2e-4, what=k)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: edge-guided attention + GRFB ablation at 3x1024x1024
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_state(module, prefix="m"):
    return {f"{prefix}.{k}": (v.detach().cpu().clone().requires_grad_(v.is_floating_point() and "running_" not in k))
            for k, v in module.state_dict().items()}


def _block_vs_oracle(module, oracle_fn, inputs, cout, out_tol, grad_rel):
    """fp32 HIP block (train mode, forward + backward) against the CPU oracle on the same seeded tensors."""
    from egm_unet_amd import ops
    st = _oracle_state(module)
    xr = [x.clone().requires_grad_(True) for x in inputs]
    ref = oracle_fn(st, *xr)
    g = torch.Generator().manual_seed(99)
    gout = torch.randn(ref.shape, generator=g) / ref.shape[1]
    ref.backward(gout)
    module.to(DEV).train()
    xs = [x.to(DEV).requires_grad_(True) for x in inputs]
    out = ops.to_nchw(module(*[ops.to_nhwc(x, torch.float32) for x in xs]), cout)
    assert_close(out.detach().cpu(), ref.detach(), what="out", **out_tol)
    out.backward(gout.to(DEV))
    from helpers import rel_err
    bad = [i for i, (x, r) in enumerate(zip(xs, xr)) if rel_err(x.grad.cpu(), r.grad) >= grad_rel]
    if bad:
        # batch-1 BatchNorm over 10^6 pixels: the fp32 oracle itself carries rounding noise of this size.  Judge against the oracle
        # run in float64: the HIP fp32 path must be as close to that truth as the reference's own fp32 arithmetic is.
        st64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone()) for k, v in st.items()}
        x64 = [x.double().clone().requires_grad_(True) for x in inputs]
        oracle_fn(st64, *x64).backward(gout.double())
        for i in bad:
            e_hip, e_ref = rel_err(xs[i].grad.cpu(), x64[i].grad), rel_err(xr[i].grad, x64[i].grad)
            assert e_hip < max(grad_rel, 2.0 * e_ref), ("input grad vs float64 oracle", i, e_hip, e_ref)
    # parameter gradients: relative to each tensor's own norm, with an absolute floor tied to the typical gradient size (the beta of
    # a BatchNorm that feeds conv -> train-mode BatchNorm has an analytically ~zero gradient: pure rounding noise in both paths)
    refs = {k: st["m." + k].grad for k, _ in module.named_parameters() if st["m." + k].grad is not None}
    floor = 1e-3 * float(torch.stack([r.double().pow(2).mean().sqrt() for r in refs.values()]).median())
    params = dict(module.named_parameters())
    worst = max((float((params[k].grad.cpu().double() - r.double()).norm()) / (float(r.double().norm()) + floor * r.numel() ** 0.5), k)
                for k, r in refs.items())
    assert worst[0] < 5 * grad_rel, worst


def test_config5_down_block_at_1024_vs_oracle():
    """Down(32, 64) fed from a 1x32x1024x1024 map: MaxPool -> conv/BN/ReLU -> MCALayer -> conv/BN/ReLU -> EdgeEnhancedGRFB at 512x512
    (K3, K1/K2, K4, K5-K8 at the ablation resolution), forward and backward, fp32 path vs the oracle."""
    from egm_unet_amd.egm_unet import Down
    from oracle import egm_ref as R
    torch.manual_seed(11)
    m = Down(32, 64)
    x = torch.randn(1, 32, 1024, 1024, generator=torch.Generator().manual_seed(12))
    _block_vs_oracle(m, lambda st, a: R.egm_down(st, "m", a, True), [x], 64, dict(rtol=2e-3, atol=3e-4), 2e-3)


def test_config5_up_block_at_1024_vs_oracle():
    """Up(64, 32): bilinear x2 of a 512x512 map + concat with the 1024x1024 skip (K10, the upsample/concat stress) -> DoubleConv."""
    from egm_unet_amd.unet import Up
    from oracle import egm_ref as R
    torch.manual_seed(13)
    m = Up(64, 32, bilinear=True)
    g = torch.Generator().manual_seed(14)
    low, skip = torch.randn(1, 32, 512, 512, generator=g), torch.randn(1, 32, 1024, 1024, generator=g)
    _block_vs_oracle(m, lambda st, a, b: R.up_block(st, "m", a, b, True), [low, skip], 32, dict(rtol=1e-3, atol=1e-4), 1e-3)


def test_config5_full_model_step_at_1024(big_model):
    """EGM-UNet(3,2,32) train step on 2x3x1024x1024: bitwise run-to-run determinism, finite gradients, bf16 tracks fp32."""
    import copy
    x, t = synth(2, 1024, 1024, 21)
    x, t = x.to(DEV), t.to(DEV)
    sd = copy.deepcopy(big_model.state_dict())
    o1, l1, g1 = _train_step_outputs(big_model, x, t, torch.bfloat16)
    big_model.load_state_dict(sd)
    o2, l2, g2 = _train_step_outputs(big_model, x, t, torch.bfloat16)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    assert all(torch.equal(g1[k], g2[k]) for k in g1)
    assert all(torch.isfinite(v).all() for v in g1.values())
    big_model.load_state_dict(sd)
    o3, l3, _ = _train_step_outputs(big_model, x, t, torch.float32)
    rel = float((o1 - o3).norm() / o3.norm())
    agree = float((o1.argmax(1) == o3.argmax(1)).float().mean())
    assert rel < 0.1 and agree > 0.97, (rel, agree)
    assert abs(float(l1) - float(l3)) < 2e-2 * abs(float(l3))
    big_model.load_state_dict(sd)
