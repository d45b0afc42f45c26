"""CPU: the oracle (oracle/*.py) against fixtures captured from the reference itself
(tools/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import egm_ref as R
from oracle import loss_ref as L
from helpers import GOLDEN, assert_close, fixture_state, load_fixture

TOL = dict(rtol=2e-4, atol=2e-5)


def _run_block(name, fn, n_in=1, gtol=None):
    fx = load_fixture(name)
    st = fixture_state(fx)
    xs = [torch.from_numpy(fx[f"in{i}"]).requires_grad_(True) for i in range(n_in)]
    out = fn(st, *xs)
    if isinstance(out, dict):
        out = out["out"]
    assert_close(out.detach(), fx["out"], what=name + " out", **TOL)
    (out * torch.from_numpy(fx["gout"])).sum().backward()
    gtol = gtol or dict(rtol=1e-3, atol=1e-4)
    for i, x in enumerate(xs):
        assert_close(x.grad, fx[f"gin{i}"], what=f"{name} gin{i}", **gtol)
    n_g = 0
    for k, v in fx.items():
        if k.startswith("grad/"):
            g = st["m." + k[5:]].grad
            assert g is not None, k
            assert_close(g, v, what=name + " " + k, **gtol)
            n_g += 1
        if k.startswith("post/"):
            assert_close(st["m." + k[5:]], v, what=name + " " + k, **TOL)
    return n_g


def test_double_conv():
    assert _run_block("double_conv", lambda st, x: R.double_conv(st, "m", x, True)) == 6


@pytest.mark.parametrize("name", ["up_block_convT", "up_block_convT_pad"])
def test_up_block_conv_transpose(name):
    _run_block(name, lambda st, a, b: R.up_block_convT(st, "m", a, b, True), n_in=2)


@pytest.mark.parametrize("name", ["up_block", "up_block_pad"])
def test_up_block(name):
    _run_block(name, lambda st, a, b: R.up_block(st, "m", a, b, True), n_in=2)


@pytest.mark.parametrize("c", [16, 64, 256])
def test_mca_layer_fft_identity(c):
    # fixture was produced by the reference's literal FFT path; the oracle uses 1.1*x
    _run_block(f"mca_c{c}", lambda st, x: R.mca_layer(st, "m", x))
    _run_block(f"mca_c{c}", lambda st, x: R.mca_layer(st, "m", x, fft_exact=True))


@pytest.mark.parametrize("c", [16, 64])
def test_mca_layer_no_spatial(c):
    # reference MCALayer(c, no_spatial=True): two gates, no c_hw parameters (tools/make_golden_mca_nospatial.py)
    _run_block(f"mca_nospatial_c{c}", lambda st, x: R.mca_layer(st, "m", x, no_spatial=True))
    _run_block(f"mca_nospatial_c{c}", lambda st, x: R.mca_layer(st, "m", x, fft_exact=True, no_spatial=True))


def test_edge_gate():
    _run_block("edge_gate", lambda st, x: R.edge_gate(st, "m", x, True))


def test_fusion_conv():
    _run_block("fusion_conv", lambda st, a, b: R.fusion_conv(st, "m", a, b), n_in=2)


@pytest.mark.parametrize("name", ["edge_grfb_c64", "edge_grfb_c32"])
def test_edge_grfb(name):
    _run_block(name, lambda st, x: R.edge_grfb(st, "m", x, True))


def test_rga():
    _run_block("rga_d64", lambda st, x: R.rga(st, "m", x))


def test_egm_down():
    # reference Down is Sequential(pool, DoubleConv1) -> keys "1.0.weight", ... under prefix m
    _run_block("egm_down", lambda st, x: R.egm_down(st, "m", x, True), gtol=dict(rtol=2e-3, atol=2e-4))


def test_unet_small():
    fx = load_fixture("unet_b8")
    st = fixture_state(fx, prefix="")
    x = torch.from_numpy(fx["in0"]).requires_grad_(True)
    out = R.unet_forward(st, x, True)["out"]
    assert_close(out.detach(), fx["out"], what="unet out", **TOL)
    (out * torch.from_numpy(fx["gout"])).sum().backward()
    for k, v in fx.items():
        if k.startswith("grad/"):
            assert_close(st[k[5:]].grad, v, rtol=2e-3, atol=2e-4, what=k)


def test_egm_unet_small_train_and_eval():
    fx = load_fixture("egm_unet_b8")
    st = fixture_state(fx, prefix="")
    x = torch.from_numpy(fx["in0"]).requires_grad_(True)
    out = R.egm_unet_forward(st, x, True)["out"]
    assert_close(out.detach(), fx["out"], what="egm out", **TOL)
    (out * torch.from_numpy(fx["gout"])).sum().backward()
    n = 0
    for k, v in fx.items():
        if k.startswith("grad/"):
            assert_close(st[k[5:]].grad, v, rtol=5e-3, atol=5e-4, what=k)
            n += 1
        if k.startswith("post/"):
            assert_close(st[k[5:]], v, what=k, **TOL)
    assert n == 333
    fe = load_fixture("egm_unet_b8_eval")
    ste = fixture_state(fe, prefix="", requires_grad=False)
    with torch.no_grad():
        oe = R.egm_unet_forward(ste, torch.from_numpy(fe["x"]), train=False)["out"]
    assert_close(oe, fe["out"], what="egm eval out", **TOL)
    assert torch.equal(oe.argmax(1), torch.from_numpy(fe["out"]).argmax(1))


@pytest.mark.parametrize("name", ["crit_small", "crit_mid"])
def test_criterion_and_metrics(name):
    fx = load_fixture(name)
    x = torch.from_numpy(fx["logits"]).requires_grad_(True)
    t = torch.from_numpy(fx["target"])
    lw = torch.tensor([1.0, 2.0])
    terms = L.criterion_terms(x, t, lw, 2, 255)
    for k, v in terms.items():
        assert_close(v.detach(), fx["term_" + k], rtol=1e-5, atol=1e-6, what=k)
    loss = L.criterion({"out": x}, t, lw, num_classes=2, ignore_index=255)
    assert_close(loss.detach(), fx["loss"], rtol=1e-5, atol=1e-5, what="loss")
    loss.backward()
    assert_close(x.grad, fx["grad"], rtol=1e-4, atol=1e-7, what="dlogits")
    cm = L.confusion_matrix(t.flatten(), x.detach().argmax(1).flatten(), 2)
    assert np.array_equal(cm.numpy(), fx["confmat"])
    ag, acc, iu = L.confusion_metrics(cm)
    assert_close(ag, fx["acc_global"], 1e-6, 1e-7); assert_close(acc, fx["acc"], 1e-6, 1e-7); assert_close(iu, fx["iu"], 1e-6, 1e-7)
    assert_close(L.eval_dice(x.detach(), t), fx["dice_metric"].reshape(()), 1e-5, 1e-6, what="dice metric")


def test_criterion_no_ignore_and_no_dice():
    fx = load_fixture("crit_noignore")
    x, t = torch.from_numpy(fx["logits"]), torch.from_numpy(fx["target"])
    assert_close(L.criterion({"out": x}, t), fx["loss"], 1e-5, 1e-5)
    assert_close(L.criterion({"out": x}, t, dice=False), fx["loss_nodice"], 1e-5, 1e-6)


def test_lr_schedule():
    lrs = load_fixture("lr_schedule")["lrs"]
    mine = [0.02 * L.lr_factor(s, 7, 5) for s in range(35)]
    assert np.allclose(mine, lrs, rtol=1e-9, atol=0)


def test_train3_trace():
    """3 SGD steps of the oracle forward + criterion + sgd_step reproduce the reference's losses and weights."""
    fx = load_fixture("train3_egm_b8")
    st = fixture_state(fx, prefix="", group="init", requires_grad=False)
    params = {k: v for k, v in st.items() if v.is_floating_point() and "running_" not in k}
    bufs = {}
    lw = torch.tensor([1.0, 2.0])
    for s in range(3):
        work = dict(st)
        for k in params:
            work[k] = params[k].detach().clone().requires_grad_(True)
        loss = L.criterion(R.egm_unet_forward(work, torch.from_numpy(fx["xs"][s]), True),
                           torch.from_numpy(fx["ts"][s]), lw, num_classes=2, ignore_index=255)
        assert abs(float(loss.detach()) - fx["losses"][s]) <= 2e-4 * abs(fx["losses"][s]), (s, float(loss.detach()), fx["losses"][s])
        loss.backward()
        grads = {k: work[k].grad for k in params}
        with torch.no_grad():
            L.sgd_step(params, grads, bufs, lr=0.02)
    st.update(params)
    for k, v in fx.items():
        if k.startswith("final/"):
            assert_close(st[k[6:]], v, rtol=2e-3, atol=2e-5, what=k)


def test_manifest_shapes_match_oracle_state_builders():
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    for tag, st in (("egm_unet_3_2_32", R.make_egm_unet_state(3, 2, 32)), ("unet_default", R.make_unet_state(1, 2, 64)),
                    ("egm_unet_3_2_8", R.make_egm_unet_state(3, 2, 8))):
        keys = man[tag]["keys"]
        assert set(keys) == set(st), (tag, set(keys) ^ set(st))
        for k, shp in keys.items():
            assert list(st[k].shape) == shp, (tag, k)
        assert len(st) == man[tag]["n_entries"]


def test_yuan_twin_without_mca():
    """Ablation twin src/yuanGRFBUNet.py (no MCALayer): same oracle with use_mca=False and shifted Sequential indices."""
    fx = load_fixture("yuan_unet_b8")
    st = fixture_state(fx, prefix="")
    x = torch.from_numpy(fx["in0"]).requires_grad_(True)
    out = R.egm_unet_forward(st, x, True, use_mca=False)["out"]
    assert_close(out.detach(), fx["out"], what="yuan out", **TOL)
    (out * torch.from_numpy(fx["gout"])).sum().backward()
    for k, v in fx.items():
        if k.startswith("grad/"):
            assert_close(st[k[5:]].grad, v, rtol=5e-3, atol=5e-4, what=k)


def test_plain_grfb_block():
    """Block-level ablation twin (src/EGM-UNet.py:977-1023)."""
    fx = load_fixture("plain_grfb_c64")
    st = fixture_state(fx, prefix="m")
    x = torch.from_numpy(fx["in0"]).requires_grad_(True)
    out = R.plain_grfb(st, "m", x, True)
    assert_close(out.detach(), fx["out"], what="plain grfb out", **TOL)
    (out * torch.from_numpy(fx["gout"])).sum().backward()
    assert_close(x.grad, fx["gin0"], rtol=5e-3, atol=5e-4, what="gin")


@pytest.mark.parametrize("name", ["ela_c64", "ela_c32_k5"])
def test_ela_block(name):
    fx = load_fixture(name)
    st = fixture_state(fx, prefix="m")
    x = torch.from_numpy(fx["in0"]).requires_grad_(True)
    out = R.ela(st, "m", x)
    assert_close(out.detach(), fx["out"], what="ela out", **TOL)
    (out * torch.from_numpy(fx["gout"])).sum().backward()
    assert_close(x.grad, fx["gin0"], rtol=2e-3, atol=1e-5, what="gin")
    for k, v in fx.items():
        if k.startswith("grad/"):
            assert_close(st["m." + k[5:]].grad, v, rtol=2e-3, atol=1e-4, what=k)


def test_hegdc_block():
    fx = load_fixture("hegdc_16_24")
    st = fixture_state(fx, prefix="m")
    x = torch.from_numpy(fx["in0"]).requires_grad_(True)
    out = R.hegdc(st, "m", x, True)
    assert_close(out.detach(), fx["out"], what="hegdc out", **TOL)
    (out * torch.from_numpy(fx["gout"])).sum().backward()
    assert_close(x.grad, fx["gin0"], rtol=5e-3, atol=5e-5, what="gin")
    for k, v in fx.items():
        if k.startswith("grad/"):
            assert_close(st["m." + k[5:]].grad, v, rtol=5e-3, atol=5e-4, what=k)
