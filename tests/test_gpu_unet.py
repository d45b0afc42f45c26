"""GPU parity of the U-Net blocks / model, the fused criterion, metrics and SGD against the golden fixtures captured
from the reference (tests/golden) and against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from helpers import assert_close, fixture_state, load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32 = dict(rtol=1e-3, atol=1e-4)          # north_star: fp32 logits within 1e-3 relative


def load_module_state(module, fx, group="state"):
    sd = {k[len(group) + 1:]: torch.from_numpy(np.array(v)) for k, v in fx.items() if k.startswith(group + "/")}
    missing = module.load_state_dict(sd, strict=True)
    return missing


def run_block(module, fx, n_in, dtype=torch.float32, grad_tol=None, out_tol=None):
    from egm_unet_amd import ops
    module.to(DEV).train()
    xs = [torch.from_numpy(fx[f"in{i}"]).to(DEV).requires_grad_(True) for i in range(n_in)]
    ys = [ops.to_nhwc(x, dtype) for x in xs]
    out = module(*ys)
    C = fx["out"].shape[1]
    out = ops.to_nchw(out, C)
    assert_close(out.detach().cpu(), fx["out"], what="out", **(out_tol or F32))
    out.backward(torch.from_numpy(fx["gout"]).to(DEV))
    gt = grad_tol or dict(rtol=2e-3, atol=2e-4)
    for i, x in enumerate(xs):
        assert_close(x.grad.cpu(), fx[f"gin{i}"], what=f"gin{i}", **gt)
    sd = module.state_dict()
    params = dict(module.named_parameters())
    for k, v in fx.items():
        if k.startswith("grad/"):
            assert params[k[5:]].grad is not None, k
            assert_close(params[k[5:]].grad.cpu(), v, what=k, **gt)
        if k.startswith("post/"):
            assert_close(sd[k[5:]].cpu(), v, what=k, rtol=1e-4, atol=1e-5)


def test_double_conv_fixture():
    from egm_unet_amd.unet import DoubleConv
    fx = load_fixture("double_conv")
    m = DoubleConv(8, 16)
    load_module_state(m, fx)
    run_block(m, fx, 1)


@pytest.mark.parametrize("name", ["up_block", "up_block_pad"])
def test_up_block_fixture(name):
    from egm_unet_amd.unet import Up
    fx = load_fixture(name)
    m = Up(32, 8, bilinear=True)
    load_module_state(m, fx)
    run_block(m, fx, 2)


@pytest.mark.parametrize("name", ["up_block_convT", "up_block_convT_pad"])
def test_up_block_conv_transpose_fixture(name):
    """Up(bilinear=False): ConvTranspose2d(2, stride 2) as 1x1 conv + pixel shuffle (+ the zero pad for odd skip sizes)."""
    from egm_unet_amd.unet import Up
    fx = load_fixture(name)
    m = Up(32, 16, bilinear=False)
    load_module_state(m, fx)
    run_block(m, fx, 2)


def test_unet_conv_transpose_state_dict_and_step():
    import json, os
    from helpers import GOLDEN
    from egm_unet_amd import UNet, GRFBUNet
    man = json.load(open(os.path.join(GOLDEN, "unet_convT_manifest.json")))
    m = UNet(3, 2, bilinear=False, base_c=8)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == man
    g = GRFBUNet(3, 2, bilinear=False, base_c=8).to(DEV).train()
    out = g(torch.randn(2, 3, 64, 64, device=DEV))["out"]
    assert out.shape == (2, 2, 64, 64)
    out.sum().backward()
    assert g.up1.up.weight.grad is not None and float(g.up1.up.weight.grad.abs().sum()) > 0


def test_unet_b8_fixture_fp32():
    from egm_unet_amd import UNet
    fx = load_fixture("unet_b8")
    m = UNet(3, 2, base_c=8)
    load_module_state(m, fx)
    m.to(DEV).train()
    x = torch.from_numpy(fx["in0"]).to(DEV)
    out = m(x)["out"]
    assert_close(out.detach().cpu(), fx["out"], what="logits", **F32)
    assert torch.equal(out.argmax(1).cpu(), torch.from_numpy(fx["out"]).argmax(1)), "argmax masks must be bit-exact (fp32 path)"
    out.backward(torch.from_numpy(fx["gout"]).to(DEV))
    params = dict(m.named_parameters())
    for k, v in fx.items():
        if k.startswith("grad/"):
            assert_close(params[k[5:]].grad.cpu(), v, what=k, rtol=5e-3, atol=5e-4)


def test_unet_b8_bf16_close_to_fp32():
    from egm_unet_amd import UNet
    fx = load_fixture("unet_b8")
    m = UNet(3, 2, base_c=8)
    load_module_state(m, fx)
    m.to(DEV).train().set_compute_dtype(torch.bfloat16)
    out = m(torch.from_numpy(fx["in0"]).to(DEV))["out"].cpu()
    ref = torch.from_numpy(fx["out"])
    rel = float((out - ref).norm() / ref.norm())
    assert rel < 5e-2, rel
    agree = float((out.argmax(1) == ref.argmax(1)).float().mean())
    assert agree > 0.97, agree


def test_unet_state_dict_and_seeded_init_match_reference():
    """Drop-in boundary: same keys/shapes as the reference and identical default init under the same seed."""
    import json, os
    from helpers import GOLDEN
    from egm_unet_amd import UNet
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))["unet_default"]
    torch.manual_seed(0)
    m = UNet()
    sd = m.state_dict()
    assert list(sd.keys()) == list(man["keys"].keys())
    for k, shp in man["keys"].items():
        assert list(sd[k].shape) == shp
    for k, s in man["init_sum"].items():
        assert abs(float(sd[k].double().sum()) - s) <= 1e-6 * max(1.0, abs(s)), k


@pytest.mark.parametrize("name", ["crit_small", "crit_mid"])
def test_criterion_fixture(name):
    from egm_unet_amd.train_utils import criterion
    from egm_unet_amd.train_utils.dice_coefficient_loss import fused_criterion
    from egm_unet_amd.train_utils.distributed_utils import ConfusionMatrix, DiceCoefficient
    fx = load_fixture(name)
    x = torch.from_numpy(fx["logits"]).to(DEV).requires_grad_(True)
    t = torch.from_numpy(fx["target"]).to(DEV)
    lw = torch.tensor([1.0, 2.0], device=DEV)
    loss, terms = fused_criterion(x, t, lw, dice=True, ignore_index=255, return_terms=True)
    for i, k in enumerate(["ce", "dice", "laplace", "lap", "sobel"], 1):
        assert_close(terms[i].cpu(), fx["term_" + k], rtol=2e-5, atol=1e-6, what=k)
    loss2 = criterion({"out": x}, t, lw, num_classes=2, ignore_index=255)
    assert_close(loss2.detach().cpu(), fx["loss"], rtol=2e-5, atol=1e-5, what="loss")
    loss2.backward()
    assert_close(x.grad.cpu(), fx["grad"], rtol=2e-4, atol=2e-7, what="dlogits")
    cm = ConfusionMatrix(2)
    cm.update_from_logits(t, x.detach())
    assert np.array_equal(cm.mat.cpu().numpy(), fx["confmat"])
    cm2 = ConfusionMatrix(2)
    cm2.update(t.flatten(), x.detach().argmax(1).flatten())
    assert np.array_equal(cm2.mat.cpu().numpy(), fx["confmat"])
    ag, acc, iu = cm.compute()
    assert_close(ag.cpu(), fx["acc_global"], 1e-6, 1e-7); assert_close(acc.cpu(), fx["acc"], 1e-6, 1e-7)
    assert_close(iu.cpu(), fx["iu"], 1e-6, 1e-7)
    dc = DiceCoefficient(num_classes=2, ignore_index=255)
    dc.update(x.detach(), t)
    assert_close(dc.value.cpu().reshape(()), fx["dice_metric"].reshape(()), 1e-5, 1e-6, what="dice metric")


def test_criterion_no_ignore_no_dice():
    from egm_unet_amd.train_utils import criterion
    fx = load_fixture("crit_noignore")
    x, t = torch.from_numpy(fx["logits"]).to(DEV), torch.from_numpy(fx["target"]).to(DEV)
    assert_close(criterion({"out": x}, t).cpu(), fx["loss"], 2e-5, 1e-5)
    assert_close(criterion({"out": x}, t, dice=False).cpu(), fx["loss_nodice"], 2e-5, 1e-6)


@pytest.mark.parametrize("nc", [3, 6])
def test_criterion_multiclass_vs_oracle(nc):
    """num_classes = 3 and 6 (the 4- and 16-class instantiations of the criterion kernels; 2 classes are the fixtures above), weights,
    ignore: against the CPU oracle on the same seeded inputs."""
    from oracle import loss_ref as L
    from egm_unet_amd.train_utils import criterion
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, nc, 24, 40, generator=g)
    t = torch.randint(0, nc, (2, 24, 40), generator=g)
    t[torch.rand(2, 24, 40, generator=g) < 0.05] = 255
    lw = torch.tensor([0.5, 1.0, 2.0, 1.5, 0.7, 1.2][:nc])
    xr = x.clone().requires_grad_(True)
    lr_ = L.criterion({"out": xr}, t, lw, num_classes=nc, ignore_index=255); lr_.backward()
    xg = x.to(DEV).requires_grad_(True)
    lg = criterion({"out": xg}, t.to(DEV), lw.to(DEV), num_classes=nc, ignore_index=255); lg.backward()
    assert_close(lg.detach().cpu(), lr_.detach(), 2e-5, 1e-5)
    assert_close(xg.grad.cpu(), xr.grad, 2e-4, 2e-7)


def test_fused_sgd_matches_torch():
    from egm_unet_amd.optim import SGD
    g = torch.Generator().manual_seed(3)
    shapes = [(7,), (3, 5, 3, 3), (1,), (129, 33)]
    ps_ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    ps_gpu = [p.detach().clone().to(DEV).requires_grad_(True) for p in ps_ref]
    o_ref = torch.optim.SGD(ps_ref, lr=0.02, momentum=0.9, weight_decay=1e-4)
    o_gpu = SGD(ps_gpu, lr=0.02, momentum=0.9, weight_decay=1e-4)
    for step in range(3):
        for pr, pg in zip(ps_ref, ps_gpu):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad = gr.clone(); pg.grad = gr.to(DEV)
        o_ref.step(); o_gpu.step()
        for pr, pg in zip(ps_ref, ps_gpu):
            assert_close(pg.detach().cpu(), pr.detach(), 1e-6, 1e-7, what=f"step {step}")


def test_unet_train_steps_vs_oracle():
    """3 SGD steps of UNet(3,2,8): product (fp32) vs the CPU oracle from identical init and data."""
    from oracle import egm_ref as R, loss_ref as L
    from egm_unet_amd import UNet
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.train_utils import criterion
    st = R.make_unet_state(3, 2, 8, seed=11)
    m = UNet(3, 2, base_c=8)
    m.load_state_dict(st, strict=True)
    m.to(DEV).train()
    opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(12)
    params = {k: v.clone() for k, v in st.items() if v.is_floating_point() and "running_" not in k}
    state = {k: v.clone() for k, v in st.items()}
    bufs = {}
    lw = torch.tensor([1.0, 2.0])
    for step in range(3):
        x = torch.randn(2, 3, 48, 64, generator=g)
        t = torch.randint(0, 2, (2, 48, 64), generator=g); t[:, :2, :] = 255
        work = dict(state)
        for k in params:
            work[k] = params[k].detach().clone().requires_grad_(True)
        lr_ = L.criterion(R.unet_forward(work, x, True), t, lw, num_classes=2, ignore_index=255)
        lr_.backward()
        with torch.no_grad():
            L.sgd_step(params, {k: work[k].grad for k in params}, bufs, lr=0.02)
        for k in state:
            if "running_" in k or "num_batches" in k:
                state[k] = work[k]
        lg = criterion(m(x.to(DEV)), t.to(DEV), lw.to(DEV), num_classes=2, ignore_index=255)
        opt.zero_grad(); lg.backward(); opt.step()
        assert abs(float(lg) - float(lr_)) <= 2e-4 * abs(float(lr_)), (step, float(lg), float(lr_))
    sd = m.state_dict()
    for k in ["in_conv.0.weight", "down4.1.3.weight", "up1.conv.0.weight", "out_conv.0.bias", "down2.1.1.running_var"]:
        ref = params[k] if k in params else state[k]
        assert_close(sd[k].cpu(), ref, rtol=2e-3, atol=2e-4, what=k)
