"""Fused BatchNorm passes against their unfused forms and against PyTorch: the conv -> BatchNorm(train) -> ReLU -> conv chain vs
PyTorch's own autograd (nn.Conv2d -> nn.BatchNorm2d -> nn.ReLU ..., src/EGM-UNet.py:44-55, 958-975), and the whole EGM-UNet train step with
and without (a) the BatchNorm + element-wise fusions (csrc/bn_fused.hip), (b) the multi-tensor BatchNorm passes of the lockstep GRFB
branches, (c) the merged launches of the branches' convolutions.  (Rounds 2-3 also carried BatchNorm-into-conv operand prologues here;
they were measured slower three times and removed in round 4)"""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


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


def _run_chain_hip(m, x_nchw, dtype):
    from egm_unet_amd import ops
    from egm_unet_amd._lib import ACT_NONE, ACT_RELU
    for p in m.parameters():
        p.grad = None
    x = x_nchw.clone().requires_grad_(True)
    h = ops.to_nhwc(x, dtype)
    h = ops.conv_bn_act(h, m.c1, m.b1, ACT_RELU)
    h = ops.conv_bn_act(h, m.c2, m.b2, ACT_NONE)
    h = ops.conv_bn_act(h, m.c3, m.b3, ACT_RELU, dil=2)
    h = ops.conv2d(h, m.c4.weight, m.c4.bias)
    out = ops.to_nchw(h, m.c4.out_channels)
    out.square().mean().backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    stats = {n: b.clone() for n, b in m.named_buffers() if "running" in n}
    return out.detach().clone(), x.grad.clone(), grads, stats


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_bn_chain_matches_torch_autograd(dtype):
    torch.manual_seed(3)
    c = 16
    ref = _Chain(c).train()
    x = torch.randn(2, c, 40, 48)
    import copy
    m = copy.deepcopy(ref).to(DEV).train()
    out_f, gx_f, g_f, st_f = _run_chain_hip(m, x.to(DEV), dtype)
    xr = x.clone().requires_grad_(True)
    o = ref(xr)
    o.square().mean().backward()

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


@pytest.mark.parametrize("which", ["elementwise", "multi", "grouped"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_fused_equals_materialised(dtype, which):
    """EGM-UNet train step: logits, loss and all 333 parameter gradients identical with and without (a) the BatchNorm + element-wise
    fusions of csrc/bn_fused.hip (EdgeAwareFeatureEnhancer gate, GRFB residual tail), (b) the multi-tensor BatchNorm passes shared by the
    lockstep GRFB branches (ops.multi_conv_bn_act) and (c) the merged launches of the branches' convolutions and data gradients
    (ops.conv_group, csrc/group.h)."""
    from egm_unet_amd import GRFBUNet, ops
    toggle = {"elementwise": ops.fuse_bn_ew, "multi": ops.fuse_bn_multi, "grouped": ops.group_convs}[which]
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
    # the fused 1x1 backward is chosen per conv by its shape and by whether a launch group is open: the two settings of "grouped" /
    # "multi" would hand some 1x1 convs to it in one run and to the kernel pair in the other (another fp32 summation order)
    c1_default = ops.fuse_c1()
    ops.fuse_c1(False)
    for fuse in (True, False):
        toggle(fuse)
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
    ops.fuse_c1(c1_default)
    (o1, l1, g1, b1), (o2, l2, g2, b2) = res
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
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
