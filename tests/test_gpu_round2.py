"""Round-2 entry points against plain PyTorch (or against the separate launches they replace, bit for bit):
launch groups (egm_group_begin/end: merged pipe / direct / weight-gradient kernels), ChannelAttentionModule.fc as one launch
(src/EGM-UNet.py:1171-1190), derived FusionConv weights packed by the deriving launch (:1210-1218), max-pool backward with the skip
gradient summed in (:908 + the U-Net skip), highpass3 of a gradient fused with the fan-in sum (:872-886)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _m():
    from egm_unet_amd import ops
    from egm_unet_amd._lib import ACT_NONE, ACT_RELU, dtype_code, lib, ptr, stream
    return ops, ACT_NONE, ACT_RELU, dtype_code, lib, ptr, stream


def _nhwc(g, N, H, W, C, dtype, scale=1.0):
    return (torch.randn(N, H, W, C, generator=g) * scale).to(DEV).to(dtype)


# three INDEPENDENT conv -> BN -> act layers of one depth, as in the GRFB branches: (Cin, Cout, k, dil, groups)
GROUP_CASES = [
    [(32, 32, 3, 12, 1), (32, 32, 3, 24, 1), (32, 32, 3, 36, 1)],       # dilated: direct kernel, one instantiation
    [(32, 32, 1, 1, 1), (32, 32, 1, 1, 1), (32, 32, 1, 1, 1)],          # 1x1 tails: pipelined kernel
    [(64, 32, 1, 1, 1), (64, 16, 1, 1, 1), (64, 16, 3, 1, 1)],          # heads: two 1x1 merge, the 3x3 goes alone
    [(16, 32, 3, 1, 16), (16, 32, 3, 1, 2)],                           # grouped 3x3 pair
]


@pytest.mark.parametrize("case", GROUP_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grouped_launches_equal_separate_launches(case, dtype):
    """ops.multi_conv_bn_act with merged launches (ops.conv_group) == the same layers launched one by one: outputs, input gradients,
    weight and BatchNorm gradients and running statistics, bit for bit."""
    ops, ACT_NONE, ACT_RELU, *_ = _m()
    g = torch.Generator().manual_seed(3)
    N, H, W = 2, 40, 64
    convs, bns, xs = [], [], []
    torch.manual_seed(1)
    for cin, cout, k, dil, groups in case:
        convs.append(nn.Conv2d(cin, cout, k, padding=dil * (k // 2), dilation=dil, groups=groups, bias=False).to(DEV))
        bns.append(nn.BatchNorm2d(cout).to(DEV).train())
        xs.append(_nhwc(g, N, H, W, cin, dtype))
    gs = [_nhwc(g, N, H, W, c[1], dtype) for c in case]
    default = ops.group_convs()
    res = []
    for grouped in (True, False):
        ops.group_convs(grouped)
        try:
            for bn in bns:
                bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.num_batches_tracked.zero_()
            for m in convs + bns:
                for p in m.parameters():
                    p.grad = None
            ins = [x.clone().requires_grad_(True) for x in xs]
            items = [(ins[k], convs[k], bns[k], ACT_RELU, case[k][3], case[k][4], None) for k in range(len(case))]
            outs = ops.multi_conv_bn_act(items)
            torch.autograd.backward(outs, gs)
            torch.cuda.synchronize()
            res.append(([o.detach().clone() for o in outs], [i.grad.clone() for i in ins],
                        [p.grad.clone() for m in convs + bns for p in m.parameters()], [bn.running_var.clone() for bn in bns]))
        finally:
            ops.group_convs(default)
    for a, b in zip(res[0], res[1]):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)


def test_group_api_errors_and_abort():
    """egm_group_end without an open group fails with a message; egm_group_abort drops a group; a second begin inside a group fails."""
    ops, *_rest, lib, ptr, stream = _m()
    L = lib()
    with pytest.raises(RuntimeError, match="no open group"):
        L.call("egm_group_end", stream())
    L.call("egm_group_begin")
    with pytest.raises(RuntimeError, match="already open"):
        L.call("egm_group_begin")
    L.cdll.egm_group_abort()
    L.call("egm_group_begin")                         # usable again
    L.call("egm_group_end", stream())                 # an empty group launches nothing


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,red", [(32, 4), (128, 4), (16, 2)])
def test_ca_mlp_matches_torch(dtype, C, red):
    """ops.ca_mlp == fc[2](relu(fc[0](pooled))) of ChannelAttentionModule (1x1 convs without bias) and its autograd."""
    ops, *_ = _m()
    g = torch.Generator().manual_seed(0)
    R, Cr = 16, C // red
    p = (torch.randn(R, 1, 1, C, generator=g)).to(DEV).to(dtype).requires_grad_(True)
    w0 = (torch.randn(Cr, C, 1, 1, generator=g) / C ** 0.5).to(DEV).requires_grad_(True)
    w2 = (torch.randn(C, Cr, 1, 1, generator=g) / Cr ** 0.5).to(DEV).requires_grad_(True)
    go = torch.randn(R, 1, 1, C, generator=g).to(DEV).to(dtype)
    out = ops.ca_mlp(p, w0, w2)
    out.backward(go)
    pr = p.detach().float().reshape(R, C).requires_grad_(True)
    w0r, w2r = w0.detach().reshape(Cr, C).clone().requires_grad_(True), w2.detach().reshape(C, Cr).clone().requires_grad_(True)
    h = torch.relu(pr @ w0r.t())
    if dtype == torch.bfloat16:
        h = h.detach().to(dtype).float() + (h - h.detach())   # the hidden activation is stored in bf16 (straight-through for the check)
    ref = h @ w2r.t()
    ref.backward(go.float().reshape(R, C))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    def close(a, b, what):
        rel = float((a.detach().float().reshape(-1) - b.detach().float().reshape(-1)).norm() / (b.detach().float().norm() + 1e-12))
        assert rel < tol, f"{what}: rel-L2 {rel:.2e}"
    close(out, ref, "logits"); close(p.grad, pr.grad, "d pooled"); close(w0.grad, w0r.grad, "d w0"); close(w2.grad, w2r.grad, "d w2")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_derived_weights_with_packs_equal_separate_pack(dtype):
    """ops.fold2 / ops.merge357 with pack_dtype: the derived fp32 weight is the same as without, and a conv through the registered packs
    gives the same result as a conv that packs the derived weight itself."""
    ops, *_ = _m()
    g = torch.Generator().manual_seed(2)
    x = _nhwc(g, 2, 24, 32, 24, dtype)
    w = torch.randn(16, 48, 1, 1, generator=g).to(DEV)
    a = ops.fold2(w)
    b = ops.fold2(w, pack_dtype=dtype)
    assert torch.equal(a, b) and torch.equal(a, w[:, :24] + w[:, 24:])
    assert torch.equal(ops.conv2d(x, a, None), ops.conv2d(x, b, None))
    ws = [torch.randn(8, 24, k, k, generator=g).to(DEV) for k in (3, 5, 7)]
    bs = [torch.randn(8, generator=g).to(DEV) for _ in range(3)]
    w7a, b7a = ops.merge357(*ws, *bs)
    w7b, b7b = ops.merge357(*ws, *bs, pack_dtype=dtype)
    ref = ws[2] + F.pad(ws[1], (1, 1, 1, 1)) + F.pad(ws[0], (2, 2, 2, 2))
    assert torch.equal(w7a, w7b) and torch.equal(b7a, b7b) and torch.allclose(w7a, ref, atol=1e-6)
    assert torch.equal(ops.conv2d(x, w7a, b7a), ops.conv2d(x, w7b, b7b))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fork_maxpool_backward_equals_pool_then_add(dtype):
    """ops.fork_maxpool2: (alias, pooled) forward; backward = maxpool backward + skip gradient, identical to the two separate passes
    and to torch's max_pool2d autograd."""
    ops, *_ = _m()
    g = torch.Generator().manual_seed(4)
    x = _nhwc(g, 2, 32, 48, 16, dtype)
    gs, gp = _nhwc(g, 2, 32, 48, 16, dtype), _nhwc(g, 2, 16, 24, 16, dtype)
    xa = x.clone().requires_grad_(True)
    s, p = ops.fork_maxpool2(xa)
    torch.autograd.backward([s, p], [gs, gp])
    xb = x.clone().requires_grad_(True)
    s2, q = ops.fork(xb, 2)
    torch.autograd.backward([s2, ops.maxpool2(q)], [gs, gp])
    assert torch.equal(p, ops.maxpool2(x)) and torch.equal(xa.grad, xb.grad)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    pr = F.max_pool2d(xr, 2)
    torch.autograd.backward([xr * 1.0, pr], [gs.float().permute(0, 3, 1, 2), gp.float().permute(0, 3, 1, 2)])
    ref = xr.grad.permute(0, 2, 3, 1)
    assert float((xa.grad.float() - ref).abs().max()) <= (1e-6 if dtype == torch.float32 else 4e-2)


@pytest.mark.parametrize("n", [1, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fork_highpass_backward_equals_separate_passes(dtype, n):
    """ops.fork_highpass3(x, n): forward highpass3(x) + n aliases; backward == highpass3(g_hp) followed by the fan-in sum, bit for bit."""
    ops, *_ = _m()
    g = torch.Generator().manual_seed(5)
    x = _nhwc(g, 2, 20, 36, 16, dtype)
    grads = [_nhwc(g, 2, 20, 36, 16, dtype) for _ in range(n + 1)]
    xa = x.clone().requires_grad_(True)
    outs = ops.fork_highpass3(xa, n)
    torch.autograd.backward(list(outs), grads)
    xb = x.clone().requires_grad_(True)
    al = ops.fork(xb, n + 1)
    torch.autograd.backward([ops.highpass3(al[0])] + list(al[1:]), grads)
    assert torch.equal(outs[0], ops.highpass3(x)) and torch.equal(xa.grad, xb.grad)
    ref = x.float() - F.avg_pool2d(x.float().permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1)
    assert float((outs[0].float() - ref).abs().max()) <= (1e-5 if dtype == torch.float32 else 4e-2)
