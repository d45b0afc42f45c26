"""The skip-connection max pool fused into its neighbours (csrc/pool_fused.hip) against the separate kernels it replaces and against
plain PyTorch: conv -> BatchNorm -> ReLU -> (skip, maxpool2) at the encoder's top level (src/EGM-UNet.py:44-55 + :908) and the
EdgeEnhancedGRFB target gate -> (skip, maxpool2) below it (:1319-1321 + :908)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _nhwc(g, N, H, W, C, dtype, scale=1.0):
    return (torch.randn(N, H, W, C, generator=g) * scale).to(DEV).to(dtype)


@pytest.mark.parametrize("shape", [(2, 32, 48, 16), (1, 6, 10, 64), (2, 16, 16, 256), (1, 4, 4, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gate3_pool_equals_gate3_then_fork_maxpool(shape, dtype):
    """ops.gate3_pool == ops.gate3 -> ops.fork_maxpool2: outputs and dx bit for bit, dt to FMA-contraction noise; also with only one of
    the two gradients present (the separate-kernel fallbacks)."""
    from egm_unet_amd import ops
    N, H, W, C = shape
    g = torch.Generator().manual_seed(11)
    x, t = _nhwc(g, N, H, W, C, dtype), _nhwc(g, N, H, W, 8, dtype)
    gs, gp = _nhwc(g, N, H, W, C, dtype), _nhwc(g, N, H // 2, W // 2, C, dtype)
    for use in ((True, True), (True, False), (False, True)):
        xa, ta = x.clone().requires_grad_(True), t.clone().requires_grad_(True)
        oa, pa = ops.gate3_pool(xa, ta)
        xb, tb = x.clone().requires_grad_(True), t.clone().requires_grad_(True)
        ob, pb = ops.fork_maxpool2(ops.gate3(xb, tb))
        assert torch.equal(oa, ob) and torch.equal(pa, pb)
        outs_a = [o for o, u in zip((oa, pa), use) if u]
        outs_b = [o for o, u in zip((ob, pb), use) if u]
        grads = [o for o, u in zip((gs, gp), use) if u]
        torch.autograd.backward(outs_a, grads)
        torch.autograd.backward(outs_b, grads)
        assert torch.equal(xa.grad, xb.grad), use
        # dt = (sum_c g*x)/3 * s(1-s): the channel dot product is contracted to FMAs at the compiler's discretion in either kernel
        terr = float((ta.grad.float() - tb.grad.float()).abs().max()) / max(1e-20, float(tb.grad.float().abs().max()))
        assert terr <= (1e-6 if dtype == torch.float32 else 8e-3), (use, terr)
        if use == (True, True):
            gx_both, gt_both = xa.grad.clone(), ta.grad.clone()
    # against torch
    xr = x.float().requires_grad_(True); tr = t.float().requires_grad_(True)
    o = xr * (1 + torch.sigmoid(tr[..., :3]).mean(-1, keepdim=True))
    p = F.max_pool2d(o.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    torch.autograd.backward([o, p], [gs.float(), gp.float()])
    tol = 1e-5 if dtype == torch.float32 else 6e-2
    assert float((oa.float() - o).abs().max()) <= tol * max(1.0, float(o.abs().max()))
    if dtype == torch.float32:
        assert float((gx_both - xr.grad).abs().max()) <= 1e-4
        assert float((gt_both[..., :3] - tr.grad[..., :3]).abs().max()) <= 1e-3 * max(1.0, float(tr.grad.abs().max()))


@pytest.mark.parametrize("shape", [(2, 32, 48, 16, 32), (1, 8, 12, 8, 16), (2, 16, 16, 32, 128)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_bn_act_pool_equals_conv_bn_act_then_fork_maxpool(shape, dtype):
    """ops.conv_bn_act_pool == ops.conv_bn_act -> ops.fork_maxpool2: z, pooled and running statistics bit for bit; dx / dW / dgamma /
    dbeta to fp32 summation-order noise (the BatchNorm partial sums are accumulated window by window instead of pixel by pixel)."""
    from egm_unet_amd import ops
    from egm_unet_amd._lib import ACT_RELU
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(12)
    torch.manual_seed(2)
    conv = nn.Conv2d(Cin, Cout, 3, padding=1, bias=False).to(DEV)
    bn = nn.BatchNorm2d(Cout).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
    x = _nhwc(g, N, H, W, Cin, dtype)
    gs, gp = _nhwc(g, N, H, W, Cout, dtype), _nhwc(g, N, H // 2, W // 2, Cout, dtype)
    res = []
    for fused in (True, False):
        bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
        for p in list(conv.parameters()) + list(bn.parameters()):
            p.grad = None
        xa = x.clone().requires_grad_(True)
        if fused:
            assert ops.pool_fusable(xa)
            z, p = ops.conv_bn_act_pool(xa, conv, bn, ACT_RELU)
        else:
            z, p = ops.fork_maxpool2(ops.conv_bn_act(xa, conv, bn, ACT_RELU))
        torch.autograd.backward([z, p], [gs, gp])
        torch.cuda.synchronize()
        res.append((z.clone(), p.clone(), xa.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                    bn.running_mean.clone(), bn.running_var.clone()))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[6], b[6]) and torch.equal(a[7], b[7])
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for k, name in ((2, "dx"), (3, "dW"), (4, "dgamma"), (5, "dbeta")):
        err = float((a[k].float() - b[k].float()).norm() / (b[k].float().norm() + 1e-20))
        assert err <= tol, (name, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_with_fused_pool_equals_separate_pool(dtype):
    """EGM-UNet(3, 2, 8) one train step with ops.fuse_pool on and off: identical logits, gradients to summation-order noise."""
    from egm_unet_amd import GRFBUNet, ops
    from oracle import egm_ref as R
    st = R.make_egm_unet_state(3, 2, 8, seed=7)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    gl = torch.randn(2, 2, 64, 64, generator=g).to(DEV)
    default = ops.fuse_pool()
    outs = []
    try:
        for fused in (True, False):
            ops.fuse_pool(fused)
            m = GRFBUNet(3, 2, base_c=8)
            m.load_state_dict(st, strict=True)
            m.to(DEV).train().set_compute_dtype(dtype)
            out = m(x)["out"]
            out.backward(gl)
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    finally:
        ops.fuse_pool(default)
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys()
    worst = 0.0
    for k in outs[0][1]:
        a, b = outs[0][1][k].float(), outs[1][1][k].float()
        if float(b.norm()) > 1e-6:
            worst = max(worst, float((a - b).norm() / b.norm()))
    assert worst <= (1e-3 if dtype == torch.float32 else 0.2), worst


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_with_mca_statistics_pass_applying_batchnorm_is_identical(dtype):
    """DoubleConv1's first BatchNorm+ReLU applied by the MCALayer's statistics pass (egm_mca_reduce_bn, ops.fuse_mca_bn) against the
    separate apply pass: same values summed in the same order, so logits and every gradient are bit-identical."""
    from egm_unet_amd import GRFBUNet, ops
    from oracle import egm_ref as R
    st = R.make_egm_unet_state(3, 2, 8, seed=9)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    gl = torch.randn(2, 2, 64, 64, generator=g).to(DEV)
    default = ops.fuse_mca_bn()
    outs = []
    try:
        for fused in (True, False):
            ops.fuse_mca_bn(fused)
            m = GRFBUNet(3, 2, base_c=8)
            m.load_state_dict(st, strict=True)
            m.to(DEV).train().set_compute_dtype(dtype)
            out = m(x)["out"]
            out.backward(gl)
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                         {k: v.clone() for k, v in m.state_dict().items() if "running" in k}))
    finally:
        ops.fuse_mca_bn(default)
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys()
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k


@pytest.mark.parametrize("model_kind", ["egm", "unet"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_with_deferred_dz_matches_separate_kernels(dtype, model_kind):
    """BatchNorm backward computing dz from the classifier's / the MCALayer's inputs (csrc/bn_dz_fused.hip, ops.fuse_dz) against the
    separate data-gradient conv / egm_mca_bwd_dx kernels: identical logits, every gradient to rounding / summation-order noise."""
    from egm_unet_amd import GRFBUNet, UNet, ops
    from oracle import egm_ref as R
    g = torch.Generator().manual_seed(14)
    x = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    gl = torch.randn(2, 2, 64, 64, generator=g).to(DEV)
    st = R.make_egm_unet_state(3, 2, 8, seed=13) if model_kind == "egm" else None
    default = ops.fuse_dz()
    outs = []
    try:
        for fused in (True, False):
            ops.fuse_dz(fused)
            torch.manual_seed(3)
            m = GRFBUNet(3, 2, base_c=8) if model_kind == "egm" else UNet(3, 2, base_c=8)
            if st is not None:
                m.load_state_dict(st, strict=True)
            m.to(DEV).train().set_compute_dtype(dtype)
            out = m(x)["out"]
            out.backward(gl)
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    finally:
        ops.fuse_dz(default)
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].keys() == outs[1][1].keys()
    worst, big = 0.0, max(float(v.float().norm()) for v in outs[1][1].values())
    for k in outs[0][1]:
        a, b = outs[0][1][k].float(), outs[1][1][k].float()
        if float(b.norm()) > 1e-5 * big:
            worst = max(worst, float((a - b).norm() / b.norm()))
    assert worst <= (2e-4 if dtype == torch.float32 else 0.1), worst


@pytest.mark.parametrize("model_kind", ["egm", "unet"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_with_classifier_in_batchnorm_apply_pass_matches_separate_conv(dtype, model_kind):
    """The 1x1 classifier evaluated inside up4's last BatchNorm apply pass (egm_bn_act_cls_fwd, ops.fuse_cls) against the separate
    apply pass + conv launch + layout conversion: logits to accumulation-order / rounding noise, gradients likewise."""
    from egm_unet_amd import GRFBUNet, UNet, ops
    from oracle import egm_ref as R
    g = torch.Generator().manual_seed(16)
    x = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    gl = torch.randn(2, 2, 64, 64, generator=g).to(DEV)
    st = R.make_egm_unet_state(3, 2, 8, seed=15) if model_kind == "egm" else None
    default = ops.fuse_cls()
    outs = []
    try:
        for fused in (True, False):
            ops.fuse_cls(fused)
            torch.manual_seed(4)
            m = GRFBUNet(3, 2, base_c=8) if model_kind == "egm" else UNet(3, 2, base_c=8)
            if st is not None:
                m.load_state_dict(st, strict=True)
            m.to(DEV).train().set_compute_dtype(dtype)
            out = m(x)["out"]
            assert out.shape == (2, 2, 64, 64) and out.dtype == torch.float32 and out.is_contiguous()
            out.backward(gl)
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
            with torch.no_grad():
                m.eval()
                ev = m(x)["out"]
                assert ev.shape == (2, 2, 64, 64) and bool(torch.isfinite(ev).all())
    finally:
        ops.fuse_cls(default)
    a, b = outs[0][0], outs[1][0]
    lerr = float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))
    assert lerr <= (1e-5 if dtype == torch.float32 else 1e-2), lerr
    assert outs[0][1].keys() == outs[1][1].keys()
    worst, big = 0.0, max(float(v.float().norm()) for v in outs[1][1].values())
    for k in outs[0][1]:
        p, q = outs[0][1][k].float(), outs[1][1][k].float()
        if float(q.norm()) > 1e-5 * big:
            worst = max(worst, float((p - q).norm() / q.norm()))
    assert worst <= (2e-4 if dtype == torch.float32 else 0.1), worst
