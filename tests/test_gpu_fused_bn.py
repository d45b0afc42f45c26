"""BatchNorm folded into its consumer (operand prologues, ops.Lazy, _ConvBN): the fused path must equal the materialised one
BIT FOR BIT (both run the same element formulas, csrc/prologue.h), and the chain must match PyTorch's own autograd
(nn.Conv2d -> nn.BatchNorm2d(train) -> nn.ReLU -> nn.Conv2d ..., src/EGM-UNet.py:44-55, 958-975)."""
import ctypes

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mods():
    from egm_unet_amd import ops
    from egm_unet_amd._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, PRE_BN_ACT, PRE_BN_BWD, PRE_NONE, dtype_code, lib, ptr, stream
    return ops, dict(NONE=ACT_NONE, RELU=ACT_RELU, SIGMOID=ACT_SIGMOID), (PRE_NONE, PRE_BN_ACT, PRE_BN_BWD), dtype_code, lib, ptr, stream


def _rand_nhwc(g, N, H, W, C, dtype, scale=1.0):
    return (torch.randn(N, H, W, C, generator=g) * scale).to(DEV).to(dtype)


def _coef(g, C, creal=None):
    creal = C if creal is None else creal
    cf = torch.zeros(4, C)
    cf[0, :creal] = torch.rand(creal, generator=g) + 0.5
    cf[1, :creal] = torch.randn(creal, generator=g) * 0.3
    cf[2, :creal] = torch.randn(creal, generator=g) * 0.2
    cf[3, :creal] = torch.rand(creal, generator=g) + 0.5
    return cf.to(DEV)


# (N, H, W, Cin, Cout, k, dil): every forward kernel family that takes the prologue
FWD_CASES = [
    (2, 40, 70, 32, 32, 3, 1),      # pipelined 3x3, ragged edge tiles
    (8, 64, 64, 32, 32, 3, 1),      # tall-tile variant (R = 4)
    (2, 48, 64, 64, 64, 3, 1),      # 64-cout tiles, two channel chunks
    (2, 33, 37, 24, 40, 3, 1),      # ragged last chunk (Cin = 24), Cout not a multiple of 32
    (2, 32, 64, 16, 16, 1, 1),      # pipelined 1x1
    (2, 32, 64, 64, 16, 1, 1),      # direct 1x1 (wide in, narrow out)
    (2, 64, 64, 16, 16, 3, 12),     # direct dilated 3x3
    (1, 40, 48, 16, 16, 7, 1),      # 7x7: prologue runs on the generic kernel
]


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("actname", ["RELU", "NONE", "SIGMOID"])
def test_conv_prologue_equals_materialised(case, dtype, actname):
    ops, ACT, (PN, PA, PB), dtype_code, lib, ptr, stream = _mods()
    N, H, W, Cin, Cout, k, dil = case
    g = torch.Generator().manual_seed(sum(case) + len(actname))
    act = ACT[actname]
    y_raw = _rand_nhwc(g, N, H, W, Cin, dtype)
    cf = _coef(g, Cin)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(DEV)
    L, dt, st = lib(), dtype_code(dtype), stream()
    wf, _ = ops._packed_weights(w, 1, dtype)
    z = torch.empty_like(y_raw)
    L.call("egm_bn_act_fwd", dt, ptr(y_raw), Cin, ptr(cf[0]), ptr(cf[1]), act, ptr(z), Cin, N * H * W, Cin, st)
    outs = []
    for pre, src in ((PN, z), (PA, y_raw)):
        nt = L.query("egm_conv_stats_tiles_pre", dt, pre, N, H, W, Cin, Cout, k, k, dil)
        stats = torch.zeros(nt, 2, Cout, device=DEV)
        o = torch.empty(N, H, W, Cout, dtype=dtype, device=DEV)
        L.call("egm_conv_fwd_pre", dt, ptr(src), Cin, pre, act, ptr(cf) if pre else None, None, 0, ptr(wf), None, 0, ptr(o), Cout, ptr(stats),
               N, H, W, Cin, Cout, k, k, dil, st)
        outs.append((o, stats.double().sum(0)))
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0][0].float()).all()
    if k == 7:      # with a prologue the 7x7 runs on the generic kernel (another summation order): equal to rounding, not bit for bit
        assert torch.allclose(outs[0][0].float(), outs[1][0].float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=2e-2 if dtype == torch.bfloat16 else 1e-4)
    else:
        assert torch.equal(outs[0][0], outs[1][0]), f"fused conv differs from conv(materialised): max {(outs[0][0].float() - outs[1][0].float()).abs().max()}"
    # the per-tile partial statistics may be tiled differently (7x7 runs another kernel with the prologue); their totals agree
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-3)


WGRAD_CASES = [
    # N, H, W, Cin, Cout, k, dil, groups
    (2, 40, 70, 32, 32, 3, 1, 1),
    (2, 48, 64, 64, 64, 3, 1, 1),     # 2 x 2 block layer (the materialised form takes the LDS-DMA path)
    (2, 33, 37, 24, 40, 3, 1, 1),
    (2, 32, 64, 64, 16, 1, 1, 1),
    (8, 128, 128, 16, 16, 3, 12, 1),  # dilated rows in one patch
    (2, 32, 32, 16, 16, 3, 24, 1),    # dilated, tap by tap
    (2, 24, 40, 8, 16, 3, 1, 2),      # grouped
]


@pytest.mark.parametrize("case", WGRAD_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("actname", ["RELU", "SIGMOID"])
def test_wgrad_prologue_equals_materialised(case, dtype, actname):
    """dW from logical operands == dW from materialised ones; the dy by-product == the stand-alone BatchNorm-backward apply."""
    ops, ACT, (PN, PA, PB), dtype_code, lib, ptr, stream = _mods()
    N, H, W, Cin, Cout, k, dil, groups = case
    g = torch.Generator().manual_seed(7)
    act = ACT[actname]
    L, dt, st = lib(), dtype_code(dtype), stream()
    CinP, CoutP = ops.pad8(Cin), ops.pad8(Cout)
    x_raw = _rand_nhwc(g, N, H, W, CinP, dtype)
    xcf = _coef(g, CinP, Cin)
    dz = _rand_nhwc(g, N, H, W, CoutP, dtype, 0.1)
    y = _rand_nhwc(g, N, H, W, CoutP, dtype)
    coef = _coef(g, CoutP, Cout)
    npix = N * H * W
    # materialised operands
    x = torch.empty_like(x_raw)
    L.call("egm_bn_act_fwd", dt, ptr(x_raw), CinP, ptr(xcf[0]), ptr(xcf[1]), ACT["RELU"], ptr(x), CinP, npix, CinP, st)
    nb = L.query("egm_channel_partials_blocks", npix, CoutP)
    part = torch.empty(nb * 2 * CoutP, device=DEV)
    L.call("egm_bn_act_bwd_reduce", dt, ptr(dz), CoutP, ptr(y), CoutP, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), act, ptr(part),
           npix, CoutP, st)
    sums_a = torch.empty(2, CoutP, device=DEV)
    L.call("egm_reduce_tiles", ptr(part), nb, CoutP, ptr(sums_a), st)
    dy = torch.empty_like(dz)
    L.call("egm_bn_act_bwd_apply", dt, ptr(dz), CoutP, ptr(y), CoutP, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), act, 1, ptr(sums_a),
           ptr(dy), CoutP, npix, CoutP, st)
    sums_b, cf4 = torch.empty(2, CoutP, device=DEV), torch.empty(4, CoutP, device=DEV)
    L.call("egm_bn_bwd_coefs", ptr(part), nb, npix, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), 1, ptr(sums_b), ptr(cf4), CoutP, st)
    assert torch.equal(sums_a, sums_b)
    ws = torch.empty(L.query("egm_conv_wgrad_workspace", N, H, W, CinP, CoutP, k, k) // 4 + 4, device=DEV)
    gw_a = torch.empty(Cout, Cin // groups, k, k, device=DEV)
    gw_b = torch.empty_like(gw_a)
    L.call("egm_conv_wgrad", dt, ptr(x), CinP, ptr(dy), CoutP, ptr(gw_a), ptr(ws), N, H, W, CinP, CoutP, Cin, Cout, k, k, dil, groups, 0, st)
    dy_out = torch.full_like(dz, float("nan"))
    L.call("egm_conv_wgrad_pre", dt, ptr(x_raw), CinP, PA, ACT["RELU"], ptr(xcf), ptr(dz), CoutP, PB, act, ptr(cf4), ptr(y), CoutP,
           ptr(dy_out), CoutP, ptr(gw_b), ptr(ws), N, H, W, CinP, CoutP, Cin, Cout, k, k, dil, groups, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(dy_out, dy), "dy by-product differs from egm_bn_act_bwd_apply"
    assert torch.isfinite(gw_a).all()
    if dtype == torch.bfloat16 and ((k in (1, 3) and (k == 1 or dil > 1)) or actname == "SIGMOID"):
        # 1- and 3-tap bf16 layers: with prologues the wave-specialised kernel runs (one workgroup per CU), without them the 4-wave
        # kernel (two per CU): another split count, i.e. another fp32 summation order of the same products.  Smooth activations: the
        # prologue form takes the 4-wave kernel (row-major k order), the materialised 3x3 form the wave-specialised one, whose
        # row-rotation loop walks the two k-step halves of a row as separate passes
        assert float((gw_a - gw_b).abs().max()) <= 2e-6 * float(gw_a.abs().max()), (float((gw_a - gw_b).abs().max()), float(gw_a.abs().max()))
    else:
        assert torch.equal(gw_a, gw_b), f"fused weight gradient differs: max {(gw_a - gw_b).abs().max():.3e} of {gw_a.abs().max():.3e}"


class _Chain(nn.Module):
    """conv3x3-BN-ReLU -> conv1x1-BN -> dilated conv3x3-BN-ReLU -> conv3x3 (+bias): every prologue consumer in one chain"""

    def __init__(self, c):
        super().__init__()
        self.c1, self.b1 = nn.Conv2d(c, c, 3, padding=1, bias=False), nn.BatchNorm2d(c)
        self.c2, self.b2 = nn.Conv2d(c, 2 * c, 1, bias=False), nn.BatchNorm2d(2 * c, momentum=0.01)
        self.c3, self.b3 = nn.Conv2d(2 * c, 2 * c, 3, padding=2, dilation=2, bias=False), nn.BatchNorm2d(2 * c)
        self.c4 = nn.Conv2d(2 * c, c, 3, padding=1)

    def forward(self, x):
        x = torch.relu(self.b1(self.c1(x)))
        x = self.b2(self.c2(x))
        x = torch.relu(self.b3(self.c3(x)))
        return self.c4(x)


def _run_chain_hip(m, x_nchw, dtype, fuse):
    from egm_unet_amd import ops
    from egm_unet_amd._lib import ACT_NONE, ACT_RELU
    default = ops.fuse_bn()
    ops.fuse_bn(fuse)
    try:
        for p in m.parameters():
            p.grad = None
        x = x_nchw.clone().requires_grad_(True)
        h = ops.to_nhwc(x, dtype)
        h = ops.conv_bn_act(h, m.c1, m.b1, ACT_RELU, lazy=True)
        h = ops.conv_bn_act(h, m.c2, m.b2, ACT_NONE, lazy=True)
        h = ops.conv_bn_act(h, m.c3, m.b3, ACT_RELU, dil=2, lazy=True)
        h = ops.conv2d(h, m.c4.weight, m.c4.bias)
        out = ops.to_nchw(h, m.c4.out_channels)
        out.square().mean().backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.clone() for n, p in m.named_parameters()}
        stats = {n: b.clone() for n, b in m.named_buffers() if "running" in n}
        return out.detach().clone(), x.grad.clone(), grads, stats
    finally:
        ops.fuse_bn(default)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chain_fused_equals_materialised_and_torch(dtype):
    torch.manual_seed(3)
    c = 16
    ref = _Chain(c).train()
    x = torch.randn(2, c, 40, 48)
    import copy
    m = copy.deepcopy(ref).to(DEV).train()
    state0 = copy.deepcopy(m.state_dict())
    out_f, gx_f, g_f, st_f = _run_chain_hip(m, x.to(DEV), dtype, True)
    m.load_state_dict(state0)
    out_m, gx_m, g_m, st_m = _run_chain_hip(m, x.to(DEV), dtype, False)
    # fused == materialised, bit for bit (outputs, input gradient, running statistics); parameter gradients: bit for bit in fp32,
    # to fp32 summation order in bf16 (the 1x1 / dilated weight gradients run with another split count when they carry prologues)
    assert torch.equal(out_f, out_m) and torch.equal(gx_f, gx_m)
    for n in g_f:
        if dtype == torch.float32:
            assert torch.equal(g_f[n], g_m[n]), n
        else:
            assert float((g_f[n] - g_m[n]).abs().max()) <= 2e-6 * float(g_m[n].abs().max()) + 1e-12, n
    for n in st_f:
        assert torch.equal(st_f[n], st_m[n]), n
    # and both == PyTorch's own modules and autograd (fp32 CPU)
    xr = x.clone().requires_grad_(True)
    o = ref(xr)
    o.square().mean().backward()
    rt, at = (2e-4, 2e-5) if dtype == torch.float32 else (6e-2, 2e-2)

    def close(a, b, what):
        a, b = a.double().cpu(), b.double()
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        # bf16 storage: 8 significant bits per tensor, compounded over four convs and three BatchNorms in both directions
        assert rel < (5e-5 if dtype == torch.float32 else (3e-2 if what == "output" else 0.15)), f"{what}: rel-L2 {rel:.3e}"
    close(out_f, o.detach(), "output")
    close(gx_f, xr.grad, "input gradient")
    for n, p in ref.named_parameters():
        if n.endswith("c1.bias"):
            continue
        close(g_f[n], p.grad, n)
    for n, b in ref.named_buffers():
        if "running" in n:
            close(st_f[n], b, n)


@pytest.mark.parametrize("which", ["prologue", "elementwise", "multi", "grouped"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_fused_equals_materialised(dtype, which):
    """EGM-UNet train step: logits, loss and all 333 parameter gradients identical with and without (a) the Lazy / operand-prologue
    path, (b) the BatchNorm + element-wise fusions of csrc/bn_fused.hip (EdgeAwareFeatureEnhancer gate, GRFB residual tail) and (c) the
    multi-tensor BatchNorm passes shared by the lockstep GRFB branches (ops.multi_conv_bn_act) and (d) the merged launches of the
    branches' convolutions and data gradients (ops.conv_group, csrc/group.h)."""
    from egm_unet_amd import GRFBUNet, ops
    toggle = {"prologue": ops.fuse_bn, "elementwise": ops.fuse_bn_ew, "multi": ops.fuse_bn_multi, "grouped": ops.group_convs}[which]
    default = toggle()
    from egm_unet_amd.train_utils import criterion
    torch.manual_seed(11)
    m = GRFBUNet(3, 2, base_c=16).to(DEV).train()
    m.set_compute_dtype(dtype)
    import copy
    state0 = copy.deepcopy(m.state_dict())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 96, 128, generator=g).to(DEV)
    t = torch.randint(0, 2, (2, 96, 128), generator=g).to(DEV)
    lw = torch.tensor([1.0, 2.0], device=DEV)
    res = []
    # the round-3 fusions (pool in the BatchNorm passes, dz on the fly, BatchNorm inside the MCA statistics pass) only exist on the
    # materialised path and accumulate the BatchNorm partial sums in another pixel order: switched off for the bit-for-bit comparison
    # of the operand-prologue path (they have their own on/off tests in test_gpu_pool_fused.py)
    from egm_unet_amd._lib import lib as _lib
    c7 = _lib().cdll.egm_conv_c7_mode                 # the 16-channel kernels take only prologue-free convs: their statistics sum in another order
    r3 = [(ops.fuse_pool, ops.fuse_pool()), (ops.fuse_dz, ops.fuse_dz()), (ops.fuse_mca_bn, ops.fuse_mca_bn()), (ops.fuse_cls, ops.fuse_cls()),
          (lambda v: c7(int(v)), c7(-1))]
    for fuse in (True, False):
        toggle(fuse)
        if which == "prologue":
            for f, _ in r3:
                f(False)
        try:
            m.load_state_dict(state0)
            for p in m.parameters():
                p.grad = None
            out = m(x)["out"]
            loss = criterion({"out": out}, t, lw, num_classes=2, ignore_index=255)
            loss.backward()
            torch.cuda.synchronize()
            res.append((out.detach().clone(), loss.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()},
                        {n: b.clone() for n, b in m.named_buffers()}))
        finally:
            toggle(default)
            for f, v in r3:
                f(v)
    (o1, l1, g1, b1), (o2, l2, g2, b2) = res
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    same = (lambda a, b: torch.equal(a, b)) if (dtype == torch.float32 or which != "prologue") else \
        (lambda a, b: float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12)      # bf16 prologue path: see the chain test
    bad = [n for n in g1 if not same(g1[n], g2[n])]
    assert not bad, f"{len(bad)} gradients differ between the fused and the materialised path, e.g. {bad[:5]}"
    bad = [n for n in b1 if not torch.equal(b1[n], b2[n])]
    assert not bad, f"buffers differ: {bad[:5]}"


def test_shared_conv_weight_gradient():
    """One weight used by two convolutions in one graph (the deferred slab reduction must not be taken): gradient == F.conv2d's."""
    import torch.nn.functional as F
    from egm_unet_amd import ops
    torch.manual_seed(0)
    w = (torch.randn(16, 16, 3, 3) * 0.1).to(DEV).requires_grad_(True)
    x = torch.randn(2, 16, 24, 40).to(DEV)
    h = ops.to_nhwc(x, torch.float32)
    y = ops.conv2d(ops.conv2d(h, w), w)
    ops.to_nchw(y, 16).square().sum().backward()
    torch.cuda.synchronize()
    wr = w.detach().cpu().clone().requires_grad_(True)
    F.conv2d(F.conv2d(x.cpu(), wr, padding=1), wr, padding=1).square().sum().backward()
    rel = float((w.grad.cpu() - wr.grad).norm() / wr.grad.norm())
    assert rel < 1e-4, rel
    # torch.autograd.grad() (no .grad accumulation) also gets the finished gradient
    w2 = (torch.randn(8, 16, 3, 3) * 0.1).to(DEV).requires_grad_(True)
    (gw,) = torch.autograd.grad(ops.to_nchw(ops.conv2d(h, w2), 8).square().sum(), [w2])
    torch.cuda.synchronize()
    w2r = w2.detach().cpu().clone().requires_grad_(True)
    F.conv2d(x.cpu(), w2r, padding=1).square().sum().backward()
    assert float((gw.cpu() - w2r.grad).norm() / w2r.grad.norm()) < 1e-4
