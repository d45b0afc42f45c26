"""1x1 conv -> BatchNorm -> activation (-> GATE / SAR) in the moment form (csrc/pw_bn.hip, ops.pw_conv_bn) against plain PyTorch fp32/fp64
on the same inputs and against the materialised chain it replaces (ops.fuse_pw(False)).  BasicConv(k = 1), EdgeAwareFeatureEnhancer and
the GRFB shortcut: src/EGM-UNet.py:872-886, 958-975, 1296-1317."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
ACTS = {0: lambda v: v, 1: torch.relu, 2: torch.sigmoid}


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _mk(cin, cout, g, bias=True):
    conv = nn.Conv2d(cin, cout, 1, bias=bias)
    bn = nn.BatchNorm2d(cout, momentum=0.01)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (1.0 / cin ** 0.5))
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g) * 0.3)
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    return conv, bn


def _torch_head(x_nchw, conv, bn, act, mode, p_nchw, alpha, training, dtype):
    """reference in float64 on the (storage-rounded) inputs; weights rounded to the storage type like the packed operands"""
    w = conv.weight.detach().to(dtype).double().requires_grad_(True)
    b = conv.bias.detach().double().requires_grad_(True) if conv.bias is not None else None
    gam, bet = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    rm, rv = bn.running_mean.detach().double().clone(), bn.running_var.detach().double().clone()
    y = F.conv2d(x_nchw, w, b)
    z = ACTS[act](F.batch_norm(y, rm, rv, gam, bet, training, bn.momentum, bn.eps))
    if mode == 0:
        out = p_nchw * (1 + z)
    elif mode == 1:
        out = torch.relu(alpha * p_nchw + z)
    else:
        out = z
    return out, (w, b, gam, bet, rm, rv)


CASES = [
    # (N, H, W, Cin, [(Cout, act, mode)], training)
    (2, 9, 7, 64, [(64, 2, 0)], True),                  # enhancer: sigmoid gate, ragged pixel count
    (2, 16, 16, 64, [(64, 0, 1)], True),                # shortcut: SAR
    (1, 12, 20, 64, [(16, 1, None), (8, 1, None)], True),   # two heads on one input
    (2, 8, 8, 8, [(8, 2, 0)], True),                    # branch enhancer, 8 channels
    (1, 10, 10, 16, [(16, 1, None)], True),
    (2, 6, 10, 128, [(128, 0, 1)], True),               # 128 channels (two 64-blocks)
    (1, 8, 12, 32, [(64, 1, None)], True),
    (2, 8, 8, 24, [(40, 1, None)], True),               # channel counts that are not powers of two
    (2, 9, 7, 64, [(64, 2, 0)], False),                 # frozen BatchNorm
    (1, 8, 8, 16, [(16, 1, None), (16, 0, None)], False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_pw_conv_bn_matches_torch(case, dtype):
    from egm_unet_amd import ops
    N, H, W, Cin, heads, training = case
    g = torch.Generator().manual_seed(Cin * 131 + len(heads) * 7 + H)
    x = (torch.randn(N, H, W, Cin, generator=g) * 1.3 + 0.2).to(dtype)
    if not ops._pw_ok(dtype, Cin, [ops.pad8(c) for c, _, _ in heads]):
        pytest.skip("shape outside the moment form's limits for this dtype")
    mods, ps, gos = [], [], []
    for cout, act, mode in heads:
        conv, bn = _mk(Cin, cout, g)
        (bn.train() if training else bn.eval())
        mods.append((conv, bn))
        ps.append((torch.randn(N, H, W, cout, generator=g)).to(dtype) if mode is not None else None)
        gos.append(torch.randn(N, H, W, cout, generator=g).to(dtype))
    # ---- reference
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    refs, prs = [], []
    for (conv, bn), (cout, act, mode), p in zip(mods, heads, ps):
        pr = p.double().permute(0, 3, 1, 2).clone().requires_grad_(True) if p is not None else None
        out, params = _torch_head(xr, conv, bn, act, mode, pr, 0.1, training, dtype)
        refs.append((out, params)); prs.append(pr)
    torch.autograd.backward([r[0] for r in refs], [go.double().permute(0, 3, 1, 2) for go in gos])
    # ---- HIP
    xd = x.to(DEV).requires_grad_(True)
    pds = [p.to(DEV).requires_grad_(True) if p is not None else None for p in ps]
    for conv, bn in mods:
        conv.to(DEV); bn.to(DEV)
    outs = ops.pw_conv_bn([(xd, [(conv, bn, act, mode, pd, 0.1, None) for (conv, bn), (cout, act, mode), pd in zip(mods, heads, pds)])])
    torch.autograd.backward(outs, [go.to(DEV) for go in gos])
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    tol_o, tol_g = (2e-5, 2e-4) if f32 else (1.2e-2, 3e-2)
    for k, ((conv, bn), (cout, act, mode)) in enumerate(zip(mods, heads)):
        ref, (w, b, gam, bet, rm, rv) = refs[k]
        assert _rel(outs[k].detach().permute(0, 3, 1, 2), ref.detach()) < tol_o, ("out", k)
        assert _rel(conv.weight.grad, w.grad) < tol_g, ("dW", k, _rel(conv.weight.grad, w.grad))
        assert _rel(bn.weight.grad, gam.grad) < tol_g, ("dgamma", k)
        assert _rel(bn.bias.grad, bet.grad) < tol_g, ("dbeta", k)
        if training:
            assert float(conv.bias.grad.abs().max()) == 0.0
            assert _rel(bn.running_mean, rm) < (1e-5 if f32 else 2e-3), ("running_mean", k)
            assert _rel(bn.running_var, rv) < (1e-5 if f32 else 5e-3), ("running_var", k)
        else:
            assert _rel(conv.bias.grad, b.grad) < tol_g, ("dbias", k)
        if mode is not None:
            assert _rel(pds[k].grad.permute(0, 3, 1, 2), prs[k].grad) < tol_g, ("dp", k)
    assert _rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < tol_g, ("dx", _rel(xd.grad.permute(0, 3, 1, 2), xr.grad))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pw_three_problems_in_one_launch_match_single_calls(dtype):
    """The three GRFB branch tails as one merged launch per kernel == three separate calls, bit for bit (outputs into channel slices of a
    concat buffer, as the block uses them)."""
    from egm_unet_amd import ops
    g = torch.Generator().manual_seed(3)
    N, H, W, C = 2, 10, 14, 16
    xs = [torch.randn(N, H, W, C, generator=g).to(dtype).to(DEV) for _ in range(3)]
    gos = [torch.randn(N, H, W, C, generator=g).to(dtype).to(DEV) for _ in range(3)]
    mods = [_mk(C, C, g, bias=False) for _ in range(3)]
    for conv, bn in mods:
        conv.to(DEV); bn.to(DEV).train()

    def run(merged):
        for conv, bn in mods:
            conv.zero_grad(); bn.zero_grad()
        buf, slots = ops.cat_slots(N, H, W, [C, C, C], dtype, DEV)
        xin = [x.clone().requires_grad_(True) for x in xs]
        if merged:
            outs = ops.pw_conv_bn([(xi, [(conv, bn, 1, None, None, 1.0, sl)]) for xi, (conv, bn), sl in zip(xin, mods, slots)])
        else:
            outs = [ops.pw_conv_bn([(xi, [(conv, bn, 1, None, None, 1.0, sl)])])[0] for xi, (conv, bn), sl in zip(xin, mods, slots)]
        torch.autograd.backward(outs, gos)
        torch.cuda.synchronize()
        return buf.clone(), [xi.grad.clone() for xi in xin], [conv.weight.grad.clone() for conv, _ in mods]
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0])
    for u, v in zip(a[1] + a[2], b[1] + b[2]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_edge_enhanced_grfb_moment_form_matches_materialised_chain(dtype):
    """EdgeEnhancedGRFB(64, 64) forward + backward with the moment form on (default) and off: same block, same inputs; fp32 agrees to
    summation-order noise, bf16 to the rounding of the conv outputs that the moment form no longer stores."""
    from egm_unet_amd import ops
    from egm_unet_amd.egm_unet import EdgeEnhancedGRFB
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 24, 20, 64, generator=g).to(dtype).to(DEV)
    go = torch.randn(2, 24, 20, 64, generator=g).to(dtype).to(DEV)
    torch.manual_seed(1)
    blk = EdgeEnhancedGRFB(64, 64).to(DEV).train()
    default = ops.fuse_pw()
    res = []
    try:
        for on in (True, False):
            ops.fuse_pw(on)
            blk.zero_grad()
            xi = x.clone().requires_grad_(True)
            out = blk(xi)
            out.backward(go)
            torch.cuda.synchronize()
            res.append((out.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None}))
    finally:
        ops.fuse_pw(default)
    f32 = dtype == torch.float32
    assert _rel(res[0][0], res[1][0]) < (1e-5 if f32 else 2e-2)
    assert _rel(res[0][1], res[1][1]) < (1e-4 if f32 else 8e-2)
    assert res[0][2].keys() == res[1][2].keys()
    big = max(float(v.norm()) for v in res[1][2].values())
    worst = max((_rel(res[0][2][k], res[1][2][k]), k) for k in res[0][2] if float(res[1][2][k].norm()) > 1e-4 * big)
    assert worst[0] < (2e-3 if f32 else 0.25), worst


def test_whole_model_with_the_moment_form_matches_the_reference_fixture():
    """GRFBUNet(3, 2, base_c=8), fp32, every eligible 1x1 conv -> BatchNorm chain in the moment form: the reference's logits (1e-3, argmax
    masks bit-exact) and gradients of the egm_unet_b8 fixture, at the tolerances of the materialised path."""
    from egm_unet_amd import GRFBUNet, ops
    from helpers import assert_close, load_fixture
    from test_gpu_unet import F32, load_module_state
    fx = load_fixture("egm_unet_b8")
    default, sites = ops.fuse_pw(), ops.pw_sites()
    try:
        ops.fuse_pw(True); ops.pw_sites({"ew", "heads", "tails"})
        m = GRFBUNet(3, 2, base_c=8)
        load_module_state(m, fx)
        m.to(DEV).train()
        out = m(torch.from_numpy(fx["in0"]).to(DEV))["out"]
        out.backward(torch.from_numpy(fx["gout"]).to(DEV))
        torch.cuda.synchronize()
    finally:
        ops.fuse_pw(default); ops.pw_sites(sites)
    assert_close(out.detach().cpu(), fx["out"], what="logits", **F32)
    assert torch.equal(out.argmax(1).cpu(), torch.from_numpy(fx["out"]).argmax(1))
    params = dict(m.named_parameters())
    rels = []
    for k, v in fx.items():
        if k.startswith("grad/"):
            ref = torch.from_numpy(v).double()
            if float(ref.norm()) >= 1e-5:
                rels.append((float((params[k[5:]].grad.cpu().double() - ref).norm() / ref.norm()), k))
    rels.sort(reverse=True)
    assert rels[0][0] < 2e-2 and rels[len(rels) // 2][0] < 2e-3, (rels[:4], rels[len(rels) // 2])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 9, 7, 64, 64), (2, 8, 8, 8, 8), (1, 10, 10, 16, 40), (2, 6, 10, 128, 128), (2, 16, 16, 128, 32),
                                   (2, 16, 16, 32, 128), (1, 33, 31, 24, 3), (4, 32, 32, 64, 16)])
def test_fused_1x1_backward_matches_the_two_kernel_backward_and_torch(shape, dtype):
    """egm_conv1x1_bwd (dx and the weight-gradient slabs from one pass over dy and x) against the data-gradient conv + weight-gradient
    slab kernel pair it replaces, and against torch: dx and dW to fp32 summation-order noise (bf16: equal after rounding up to rare last-bit flips)."""
    from egm_unet_amd import ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, H, W, ops.pad8(Cin), generator=g).to(dtype)
    x[..., Cin:] = 0
    go = torch.randn(N, H, W, ops.pad8(Cout), generator=g).to(dtype)
    go[..., Cout:] = 0
    w0 = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b0 = torch.randn(Cout, generator=g)
    default, maxc = ops.fuse_c1(), ops._C1_MAXC
    res = []
    try:
        ops._C1_MAXC = 128                     # the kernel's own limit (the product offers it up to 64 channels, where it wins)
        for on in (True, False):
            ops.fuse_c1(on)
            w, b = nn.Parameter(w0.clone().to(DEV)), nn.Parameter(b0.clone().to(DEV))
            xi = x.to(DEV).requires_grad_(True)
            ops.conv2d(xi, w, b).backward(go.to(DEV))
            torch.cuda.synchronize()
            res.append((xi.grad.clone(), w.grad.clone(), b.grad.clone()))
    finally:
        ops.fuse_c1(default); ops._C1_MAXC = maxc
    if dtype == torch.float32:                 # the fp32 MFMA forms walk k in another order: summation-order noise
        assert _rel(res[0][0], res[1][0]) < 1e-6
    else:                                      # same bf16 products, fp32 accumulation in another order: equal after rounding up to rare last-bit flips
        ndiff = int((res[0][0] != res[1][0]).sum())
        assert ndiff <= 2e-3 * res[0][0].numel() and _rel(res[0][0], res[1][0]) < 1e-3, ndiff
    assert torch.equal(res[0][2], res[1][2])
    assert _rel(res[0][1], res[1][1]) < 2e-6
    # torch, on the storage-rounded operands
    xr = x[..., :Cin].double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w0.to(dtype).double().requires_grad_(True)
    F.conv2d(xr, wr, b0.double()).backward(go[..., :Cout].double().permute(0, 3, 1, 2))
    f32 = dtype == torch.float32
    assert _rel(res[0][0][..., :Cin].permute(0, 3, 1, 2), xr.grad) < (1e-5 if f32 else 6e-3)
    assert _rel(res[0][1], wr.grad) < (1e-5 if f32 else 1e-5 + 2e-3 * 0)      # products of storage-rounded operands, fp32 accumulation
