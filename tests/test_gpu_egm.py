"""GPU parity of the EGM-UNet blocks and the whole network (fp32 path) against the golden fixtures captured from the
reference, plus bf16-vs-fp32 closeness and the drop-in boundary (state_dict keys, seeded init)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close, load_fixture
from test_gpu_unet import DEV, F32, load_module_state, run_block

pytestmark = pytest.mark.gpu
GT = dict(rtol=3e-3, atol=3e-4)


@pytest.mark.parametrize("c", [16, 64, 256])
def test_mca_layer_fixture(c):
    from egm_unet_amd.egm_unet import MCALayer
    fx = load_fixture(f"mca_c{c}")
    m = MCALayer(c)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=GT)


@pytest.mark.parametrize("c", [16, 64])
def test_mca_layer_no_spatial_fixture(c):
    """MCALayer(c, no_spatial=True) (src/EGM-UNet.py:700-703,766-771): two gates, x*(g_h+g_w)/2, no c_hw parameters; forward, dx and the
    four parameter gradients against the reference's own fixture (tools/make_golden_mca_nospatial.py)."""
    from egm_unet_amd.egm_unet import MCALayer
    fx = load_fixture(f"mca_nospatial_c{c}")
    m = MCALayer(c, no_spatial=True)
    assert not hasattr(m, "c_hw") and len(list(m.parameters())) == 4
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=GT)


def test_edge_gate_fixture():
    from egm_unet_amd.egm_unet import EdgeAwareFeatureEnhancer
    fx = load_fixture("edge_gate")
    m = EdgeAwareFeatureEnhancer(16)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=GT)


def test_fusion_conv_fixture_two_inputs():
    from egm_unet_amd.egm_unet import FusionConv
    fx = load_fixture("fusion_conv")
    m = FusionConv(28, 16)
    load_module_state(m, fx)
    # 28 real channels per input (not a multiple of 8): each input is padded to 32 on its own and `down.weight`'s columns follow
    run_block(m, fx, 2, grad_tol=GT)


@pytest.mark.parametrize("name,c", [("edge_grfb_c64", 64), ("edge_grfb_c32", 32)])
def test_edge_grfb_fixture(name, c):
    from egm_unet_amd.egm_unet import EdgeEnhancedGRFB
    fx = load_fixture(name)
    m = EdgeEnhancedGRFB(c, c, stride=1, scale=0.1, visual=12)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=dict(rtol=5e-3, atol=5e-4))


def test_plain_grfb_fixture():
    """Block-level ablation twin src/EGM-UNet.py:977-1023 (incl. the depthwise groups=2i conv) against its own fixture."""
    from egm_unet_amd.egm_unet import GRFB
    fx = load_fixture("plain_grfb_c64")
    m = GRFB(64, 64, stride=1, scale=0.1, visual=12)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=dict(rtol=5e-3, atol=5e-4))


def test_rga_fixture():
    from egm_unet_amd.egm_unet import RecursiveGatedAttention
    fx = load_fixture("rga_d64")
    m = RecursiveGatedAttention(64)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=GT)


def test_rga_order3_fixture():
    """RecursiveGatedAttention(128, order=3): split sizes [32, 32, 64], two transform convs -- the generic recursion of
    src/EGM-UNet.py:518-547 against the reference's own forward / backward (tools/make_golden_rga.py)."""
    from egm_unet_amd.egm_unet import RecursiveGatedAttention
    fx = load_fixture("rga_d128_o3")
    m = RecursiveGatedAttention(128, order=3)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=GT)


def test_egm_down_fixture():
    from egm_unet_amd.egm_unet import Down
    fx = load_fixture("egm_down")
    m = Down(8, 16)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=dict(rtol=5e-3, atol=5e-4))


def test_egm_unet_b8_fixture_fp32_train_and_eval():
    from egm_unet_amd import GRFBUNet
    fx = load_fixture("egm_unet_b8")
    m = GRFBUNet(3, 2, base_c=8)
    load_module_state(m, fx)
    m.to(DEV).train()
    out = m(torch.from_numpy(fx["in0"]).to(DEV))["out"]
    assert_close(out.detach().cpu(), fx["out"], what="logits", **F32)
    assert torch.equal(out.argmax(1).cpu(), torch.from_numpy(fx["out"]).argmax(1)), "argmax masks must be bit-exact (fp32 path)"
    out.backward(torch.from_numpy(fx["gout"]).to(DEV))
    params = dict(m.named_parameters())
    n = 0
    rels = []
    for k, v in fx.items():
        if k.startswith("grad/"):
            g = params[k[5:]].grad
            assert g is not None, k
            ref = torch.from_numpy(v).double()
            err = float((g.cpu().double() - ref).norm())
            # conv biases in front of a train-mode BN have an analytically zero gradient (pure rounding noise on both sides)
            if float(ref.norm()) < 1e-5:
                assert err < 1e-4, (k, err)
            else:
                rels.append((err / float(ref.norm()), k))
            n += 1
    assert n == 333
    rels.sort(reverse=True)
    print("worst gradient rel-L2 errors:", rels[:5], "median", rels[len(rels) // 2])
    assert rels[0][0] < 2e-2, rels[:5]
    assert rels[len(rels) // 2][0] < 2e-3, rels[len(rels) // 2]
    sd = m.state_dict()
    for k, v in fx.items():
        if k.startswith("post/"):
            assert_close(sd[k[5:]].cpu(), v, what=k, rtol=1e-3, atol=1e-5)
    # eval mode with the running statistics of the reference after its own train-mode forward
    fe = load_fixture("egm_unet_b8_eval")
    me = GRFBUNet(3, 2, base_c=8)
    load_module_state(me, fe)
    me.to(DEV).eval()
    with torch.no_grad():
        oe = me(torch.from_numpy(fe["x"]).to(DEV))["out"].cpu()
    assert_close(oe, fe["out"], what="eval logits", **F32)
    assert torch.equal(oe.argmax(1), torch.from_numpy(fe["out"]).argmax(1))


def test_egm_unet_b8_bf16_close_to_fp32():
    from egm_unet_amd import GRFBUNet
    fx = load_fixture("egm_unet_b8")
    m = GRFBUNet(3, 2, base_c=8)
    load_module_state(m, fx)
    m.to(DEV).train().set_compute_dtype(torch.bfloat16)
    out = m(torch.from_numpy(fx["in0"]).to(DEV))["out"].cpu()
    ref = torch.from_numpy(fx["out"])
    rel = float((out - ref).norm() / ref.norm())
    assert rel < 8e-2, rel
    assert float((out.argmax(1) == ref.argmax(1)).float().mean()) > 0.95


def test_egm_unet_b8_bf16_gradients_vs_fp32_fixture():
    """The benchmarked dtype, whole model: the parameter gradients of the bf16 path (bf16 activations / MFMA operands, fp32
    accumulation and weights) against the REFERENCE's fp32 gradients of the fixture.  bf16 keeps 8 significant bits per stored
    activation, and every flipped ReLU / max-pool / arg-max decision is an O(1) error in the gradient (tests/test_gpu_fullsize.py
    explains the arithmetic and pins the headline shape); on this 64 x 64, base_c = 8 fixture the deepest BatchNorms average 32
    values, which adds statistical noise on top, and a few gate weights with three elements can even change sign.  The bar is
    therefore aggregate (median / 90th percentile of the per-tensor rel-L2) plus exactness where no rectifier lies in between."""
    from egm_unet_amd import GRFBUNet
    fx = load_fixture("egm_unet_b8")
    m = GRFBUNet(3, 2, base_c=8)
    load_module_state(m, fx)
    m.to(DEV).train().set_compute_dtype(torch.bfloat16)
    out = m(torch.from_numpy(fx["in0"]).to(DEV))["out"]
    out.backward(torch.from_numpy(fx["gout"]).to(DEV))
    params = dict(m.named_parameters())
    rels, cosines = [], []
    gmax = max(float(np.linalg.norm(v)) for k, v in fx.items() if k.startswith("grad/"))
    for k, v in fx.items():
        if not k.startswith("grad/"):
            continue
        g = params[k[5:]].grad
        assert g is not None and torch.isfinite(g).all(), k
        ref = torch.from_numpy(v).double().flatten()
        got = g.cpu().double().flatten()
        if float(ref.norm()) < 1e-4 * gmax:              # analytically ~zero gradients (conv bias in front of a train-mode BN): noise on both sides
            assert float(got.norm()) < 1e-2 * gmax, (k, float(got.norm()))
            continue
        rels.append((float((got - ref).norm() / ref.norm()), k))
        cosines.append((float(torch.dot(got, ref) / (got.norm() * ref.norm())), k))
    rels.sort()
    cosines.sort()
    med, p90, worst = rels[len(rels) // 2][0], rels[int(len(rels) * 0.9)][0], rels[-1]
    print("bf16 whole-model gradients vs fp32 fixture: %d tensors, rel-L2 median %.4f, p90 %.4f, worst %.4f (%s); min cosine %.4f (%s)"
          % (len(rels), med, p90, worst[0], worst[1], cosines[0][0], cosines[0][1]))
    assert len(rels) > 150
    assert med < 0.6 and p90 < 0.85, (med, p90)                     # measured 0.479 / 0.687
    top = dict((k, r) for r, k in rels)
    assert top["grad/out_conv.0.weight"] < 0.08, top["grad/out_conv.0.weight"]     # measured 0.046: only the forward's bf16 error reaches it


def test_egm_unet_state_dict_and_seeded_init_match_reference():
    from egm_unet_amd import GRFBUNet
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))["egm_unet_3_2_32"]
    torch.manual_seed(0)
    m = GRFBUNet(3, 2, base_c=32)
    sd = m.state_dict()
    assert len(sd) == 555 and sum(p.numel() for p in m.parameters()) == 6302833
    assert list(sd.keys()) == list(man["keys"].keys())
    for k, shp in man["keys"].items():
        assert list(sd[k].shape) == shp, k
    for k, s in man["init_sum"].items():
        assert abs(float(sd[k].double().sum()) - s) <= 1e-6 * max(1.0, abs(s)), k


def test_egm_train3_trace_fp32():
    """3 SGD steps from the reference's seeded init on the reference's batches: losses and probe weights."""
    from egm_unet_amd import GRFBUNet
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.train_utils import criterion
    fx = load_fixture("train3_egm_b8")
    m = GRFBUNet(3, 2, base_c=8)
    load_module_state(m, fx, group="init")
    m.to(DEV).train()
    opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    lw = torch.tensor([1.0, 2.0], device=DEV)
    for s in range(3):
        loss = criterion(m(torch.from_numpy(fx["xs"][s]).to(DEV)), torch.from_numpy(fx["ts"][s]).to(DEV), lw, num_classes=2,
                         ignore_index=255)
        opt.zero_grad(); loss.backward(); opt.step()
        assert abs(float(loss.detach()) - fx["losses"][s]) <= 5e-4 * abs(fx["losses"][s]), (s, float(loss.detach()), fx["losses"][s])
    sd = m.state_dict()
    for k, v in fx.items():
        if k.startswith("final/"):
            ref = torch.from_numpy(v).double()
            rel = float((sd[k[6:]].cpu().double() - ref).norm() / (ref.norm() + 1e-12))
            print(k, "rel-L2 after 3 steps:", rel)
            assert rel < 5e-3, (k, rel)


def test_graphed_train_step_matches_reference_trace():
    """hipGraph replay of the whole step (what bench.py times) reproduces the reference's 3-step loss trace:
    step 0 runs eagerly as the capture warm-up, steps 1-2 are graph replays with new batches copied into the static buffers."""
    from egm_unet_amd import GRFBUNet
    from egm_unet_amd.graph import GraphedTrainStep
    from egm_unet_amd.optim import SGD
    fx = load_fixture("train3_egm_b8")
    m = GRFBUNet(3, 2, base_c=8)
    load_module_state(m, fx, group="init")
    m.to(DEV).train()
    opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    lw = torch.tensor([1.0, 2.0], device=DEV)
    xs, ts = torch.from_numpy(fx["xs"]).to(DEV), torch.from_numpy(fx["ts"]).to(DEV)
    step = GraphedTrainStep(m, opt, xs[0], ts[0], lw, num_classes=2, ignore_index=255, warmup=1)
    for s in (1, 2):
        loss = float(step(xs[s], ts[s]))
        assert abs(loss - fx["losses"][s]) <= 5e-4 * abs(fx["losses"][s]), (s, loss, fx["losses"][s])
    sd = m.state_dict()
    for k, v in fx.items():
        if k.startswith("final/"):
            ref = torch.from_numpy(v).double()
            rel = float((sd[k[6:]].cpu().double() - ref).norm() / (ref.norm() + 1e-12))
            assert rel < 5e-3, (k, rel)
    assert int(sd["in_conv.1.num_batches_tracked"]) == 3


def test_yuan_twin_fixture_fp32():
    """GRFBUNet(use_mca=False) == src/yuanGRFBUNet.py: state_dict keys, logits and gradients against its fixture."""
    from egm_unet_amd import GRFBUNet
    fx = load_fixture("yuan_unet_b8")
    m = GRFBUNet(3, 2, base_c=8, use_mca=False)
    load_module_state(m, fx)                                   # strict=True: identical keys to the twin's state_dict
    m.to(DEV).train()
    out = m(torch.from_numpy(fx["in0"]).to(DEV))["out"]
    assert_close(out.detach().cpu(), fx["out"], what="logits", **F32)
    out.backward(torch.from_numpy(fx["gout"]).to(DEV))
    params = dict(m.named_parameters())
    rels = []
    for k, v in fx.items():
        if k.startswith("grad/"):
            ref = torch.from_numpy(v).double()
            if float(ref.norm()) > 1e-5:
                rels.append(float((params[k[5:]].grad.cpu().double() - ref).norm() / ref.norm()))
    rels.sort()
    assert rels[len(rels) // 2] < 2e-3 and rels[-1] < 3e-2, (rels[len(rels) // 2], rels[-1])


@pytest.mark.parametrize("k,d,g", [(1, 1, 1), (3, 1, 1), (3, 2, 4)])
def test_conv_bn_silu_block_vs_torch(k, d, g):
    """Conv (conv -> BatchNorm2d -> SiLU, src/EGM-UNet.py:25-43) forward and backward against the same torch modules on the CPU."""
    from egm_unet_amd.egm_unet import Conv
    from egm_unet_amd import ops
    gen = torch.Generator().manual_seed(k * 10 + d)
    torch.manual_seed(3)
    m = Conv(16, 24 if g == 1 else 16, k, d=d, g=g)
    ref = torch.nn.Sequential(torch.nn.Conv2d(16, m.conv.out_channels, k, 1, d * (k - 1) // 2, groups=g, dilation=d, bias=False),
                              torch.nn.BatchNorm2d(m.conv.out_channels), torch.nn.SiLU())
    ref[0].load_state_dict(m.conv.state_dict()); ref[1].load_state_dict(m.bn.state_dict())
    x = torch.randn(2, 16, 20, 28, generator=gen)
    gz = torch.randn(2, m.conv.out_channels, 20, 28, generator=gen)
    xr = x.clone().requires_grad_(True)
    zr = ref(xr); zr.backward(gz)
    m.to(DEV).train()
    xg = ops.to_nhwc(x.to(DEV), torch.float32).requires_grad_(True)
    z = m(xg)
    z.backward(ops.to_nhwc(gz.to(DEV), torch.float32))
    C = m.conv.out_channels
    assert_close(ops.to_nchw(z.detach(), C).cpu(), zr.detach(), rtol=2e-4, atol=2e-5, what="z")
    assert_close(ops.to_nchw(xg.grad, 16).cpu(), xr.grad, rtol=2e-3, atol=2e-5, what="dx")
    assert_close(m.conv.weight.grad.cpu(), ref[0].weight.grad, rtol=2e-3, atol=2e-4, what="dw")
    assert_close(m.bn.weight.grad.cpu(), ref[1].weight.grad, rtol=2e-3, atol=2e-4, what="dgamma")
    assert_close(m.bn.running_var.cpu(), ref[1].running_var, rtol=1e-4, atol=1e-5, what="running_var")


@pytest.mark.parametrize("name,c,k", [("ela_c64", 64, 7), ("ela_c32_k5", 32, 5)])
def test_ela_fixture(name, c, k):
    """ELA (src/EGM-UNet.py:56-79, GroupNorm(16) strip attention) forward / input gradient / parameter gradients vs its fixture."""
    from egm_unet_amd.egm_unet import ELA
    fx = load_fixture(name)
    m = ELA(c, kernel_size=k)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=dict(rtol=2e-3, atol=2e-4))


def test_hegdc_fixture():
    """HEGDC (src/EGM-UNet.py:210-340): edge features under no_grad with batch-global min-max, sigmoid(den) weight scaling, gated
    double conv -- forward, input gradient and every parameter gradient (incl. den, alpha, edge_fusion) vs the reference fixture."""
    from egm_unet_amd.egm_unet import HEGDC
    fx = load_fixture("hegdc_16_24")
    m = HEGDC(16, 24)
    load_module_state(m, fx)
    run_block(m, fx, 1, grad_tol=dict(rtol=5e-3, atol=5e-4))


def _blob_dataset(n, size, seed):
    """Learnable synthetic segmentation data: noise + a rectangular foreground blob with a colour offset; 1 % ignore pixels."""
    g = torch.Generator().manual_seed(seed)
    x = 0.6 * torch.randn(n, 3, size, size, generator=g)
    t = torch.zeros(n, size, size, dtype=torch.int64)
    for i in range(n):
        cy, cx = torch.randint(size // 4, 3 * size // 4, (2,), generator=g).tolist()
        hh, ww = torch.randint(size // 6, size // 3, (2,), generator=g).tolist()
        t[i, max(0, cy - hh):cy + hh, max(0, cx - ww):cx + ww] = 1
    fg = (t == 1).float().unsqueeze(1)
    x = x + fg * torch.tensor([1.0, -0.6, 0.4]).view(1, 3, 1, 1)
    t[torch.rand(n, size, size, generator=g) < 0.01] = 255
    return x, t


TRAIN_RUN = (64, 4, 8, 25, 0.02, 48)          # size, batch, batches per epoch, epochs, lr0, steps compared one by one


def _train_run_data():
    from oracle import egm_ref as R
    size, bs, nb = TRAIN_RUN[:3]
    xs, ts = _blob_dataset(bs * nb, size, 31)
    xv, tv = _blob_dataset(64, size, 32)
    return xs, ts, xv, tv, R.make_egm_unet_state(3, 2, 8, seed=9), torch.tensor([1.0, 2.0])


def _oracle_eval(work):
    from oracle import egm_ref as R, loss_ref as L
    _, _, xv, tv, _, _ = _train_run_data()
    with torch.no_grad():
        pred = R.egm_unet_forward(work, xv, False)["out"].argmax(1)
    mat = L.confusion_matrix(tv.flatten(), pred.flatten(), 2)
    return mat, float(L.confusion_metrics(mat)[2].mean()) * 100


_oracle_runs_cache = []


def _oracle_training_runs():
    """The CPU oracle trained for 200 steps with the reference's recipe, twice (8 threads and 4 threads: identical arithmetic, another
    reduction order) -> ((losses, val mIoU), (losses, val mIoU)).  Run once per test session (fp32 and bf16 training tests share it)."""
    if _oracle_runs_cache:
        return _oracle_runs_cache[0]
    from oracle import egm_ref as R, loss_ref as L
    size, bs, nb, epochs, lr0, _ = TRAIN_RUN
    xs, ts, _, _, st, lw = _train_run_data()

    def oracle_run(threads):
        torch.set_num_threads(threads)
        params = {k: v.clone() for k, v in st.items() if v.is_floating_point() and "running_" not in k}
        work, bufs, losses = {k: v.clone() for k, v in st.items()}, {}, []
        for step in range(epochs * nb):
            b = step % nb
            for k in params:
                work[k] = params[k].detach().clone().requires_grad_(True)
            loss = L.criterion(R.egm_unet_forward(work, xs[b * bs:(b + 1) * bs], True), ts[b * bs:(b + 1) * bs], lw, num_classes=2,
                               ignore_index=255)
            loss.backward()
            with torch.no_grad():
                L.sgd_step(params, {k: work[k].grad for k in params}, bufs, lr=lr0 * L.lr_factor(step, nb, epochs))
            losses.append(float(loss.detach()))
        for k in params:
            work[k] = params[k].detach()
        return losses, _oracle_eval(work)[1]

    nthreads = torch.get_num_threads()
    try:
        a = oracle_run(min(8, nthreads))
        b = oracle_run(max(1, min(8, nthreads) // 2))
    finally:
        torch.set_num_threads(nthreads)
    _oracle_runs_cache.append((a, b))
    return a, b


def test_training_run_val_miou_matches_oracle():
    """SURVEY 8d mIoU check (ii): 200 training steps of the HIP build and of the CPU oracle from the same seeded init on the same
    synthetic batches with the reference's recipe (SGD 0.9 / 1e-4, warm-up + poly LR stepped per iteration, 5-term criterion,
    weights [1, 2]); held-out set = 64 seeded images.

    What is asserted, and why in this form.  Training with this criterion is chaotic at the level of fp32 summation order (its
    Laplace / Sobel terms are L1 norms: a gradient sign flips when a near-zero response changes sign).  Measured on the oracle
    ALONE: the same 200 steps with 8 and with 4 CPU threads -- identical arithmetic, another reduction order -- end with weights that
    differ by up to 5 % (relative L2 per tensor), running variances by 14 %, and held-out mIoU 97.44 vs 97.74; giving one run the
    other's BatchNorm statistics makes it worse (98.11 / 96.50), so "copy the statistics" does not make the comparison well posed,
    and neither does training longer.  The reference therefore does not agree with ITSELF to +-0.1 after training, and no
    implementation can.  The well-posed statements are:
      (a) step for step, while the trajectories still coincide, the losses agree: <= 1e-3 relative over the first 48 steps
          (measured 4e-5 .. 8e-5);
      (b) the TRAINED model is the same function in both implementations: the weights and running statistics the HIP build ends
          with, evaluated on the 64 held-out images by the oracle and by the build, give mIoU within +-0.1 (fp32 path: identical
          confusion matrix; bf16 path: within 0.1) -- north_star's bar, on a trained model;
      (c) the build's own training result lies in the band the oracle's runs span (+-0.75, the oracle's measured self-spread)."""
    from egm_unet_amd import GRFBUNet
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.train_utils import create_lr_scheduler, criterion
    from egm_unet_amd.train_utils.distributed_utils import ConfusionMatrix
    size, bs, nb, epochs, lr0, ncmp = TRAIN_RUN
    xs, ts, xv, tv, st, lw = _train_run_data()
    oracle_eval = _oracle_eval
    (ref_losses, miou_a), (ref_losses_b, miou_b) = _oracle_training_runs()
    # ---- HIP run (fp32 path)
    m = GRFBUNet(3, 2, base_c=8)
    m.load_state_dict(st, strict=True)
    m.to(DEV).train().set_compute_dtype(torch.float32)
    opt = SGD(m.parameters(), lr=lr0, momentum=0.9, weight_decay=1e-4)
    sched = create_lr_scheduler(opt, nb, epochs, warmup=True)
    lwd, losses = lw.to(DEV), []
    for step in range(epochs * nb):
        b = step % nb
        loss = criterion(m(xs[b * bs:(b + 1) * bs].to(DEV)), ts[b * bs:(b + 1) * bs].to(DEV), lwd, num_classes=2, ignore_index=255)
        opt.zero_grad(); loss.backward(); opt.step(); sched.step()
        losses.append(float(loss.detach()))
    m.eval()

    def hip_eval():
        cm = ConfusionMatrix(2)
        with torch.no_grad():
            for i in range(0, xv.shape[0], 16):
                cm.update_from_logits(tv[i:i + 16].to(DEV), m(xv[i:i + 16].to(DEV))["out"])
        return cm.mat.cpu(), float(cm.compute()[2].mean()) * 100

    mat_hip, miou = hip_eval()
    trained = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    mat_ref, miou_same_weights = oracle_eval(trained)                  # the reference's algorithm on the build's trained model
    m.set_compute_dtype(torch.bfloat16)
    _, miou_bf16 = hip_eval()
    worst = max(abs(a - b) / abs(b) for a, b in zip(losses[:ncmp], ref_losses[:ncmp]))
    self_worst = max(abs(a - b) / abs(b) for a, b in zip(ref_losses_b[:ncmp], ref_losses[:ncmp]))
    print(f"val mIoU (64 images): build-trained model by build {miou:.3f} (bf16 {miou_bf16:.3f}) / by oracle {miou_same_weights:.3f}; "
          f"oracle-trained {miou_a:.3f} / {miou_b:.3f}; loss first/last hip {losses[0]:.4f}/{losses[-1]:.4f} oracle "
          f"{ref_losses[0]:.4f}/{ref_losses[-1]:.4f}; first {ncmp} steps worst rel hip-oracle {worst:.2e}, oracle-oracle {self_worst:.2e}")
    assert ref_losses[-1] < 0.9 * ref_losses[0], "the synthetic task must be learnable for the comparison to mean anything"
    assert worst < 1e-3, worst                                                          # (a)
    assert abs(miou - miou_same_weights) <= 0.1, (miou, miou_same_weights)             # (b) fp32: +-0.1 (same confusion matrix up to logit ties)
    moved = int((mat_hip.to(torch.int64) - mat_ref.to(torch.int64)).abs().sum()) // 2   # pixels the two evaluations classify differently
    assert moved <= 1e-4 * int(mat_ref.sum()), (mat_hip, mat_ref)
    assert abs(miou_bf16 - miou_same_weights) <= 0.1, (miou_bf16, miou_same_weights)   # (b) bf16
    lo, hi = min(miou_a, miou_b), max(miou_a, miou_b)
    assert lo - 0.75 <= miou <= hi + 0.75, (miou, miou_a, miou_b)                      # (c)


def test_train_one_epoch_graph_replay_equals_eager_loop(monkeypatch):
    """train_utils.train_one_epoch: the hipGraph-replayed loop (default) and the eager loop (EGM_GRAPH_TRAIN=0) give bit-identical
    weights and the same mean loss / final lr over two epochs with a per-iteration LR schedule and a short last batch."""
    import copy
    from egm_unet_amd import GRFBUNet
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.train_utils import create_lr_scheduler, train_one_epoch
    xs, ts = _blob_dataset(14, 64, 41)
    loader = [(xs[i:i + 4], ts[i:i + 4]) for i in range(0, 14, 4)]        # 4, 4, 4, 2 images
    torch.manual_seed(3)
    sd0 = copy.deepcopy(GRFBUNet(3, 2, base_c=8).state_dict())

    def run(graph):
        monkeypatch.setenv("EGM_GRAPH_TRAIN", "1" if graph else "0")
        m = GRFBUNet(3, 2, base_c=8)
        m.load_state_dict(sd0)
        m.to(DEV)
        opt = SGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
        sched = create_lr_scheduler(opt, len(loader), 2, warmup=True)
        out = [train_one_epoch(m, opt, loader, DEV, ep, 2, sched, print_freq=100) for ep in range(2)]
        torch.cuda.synchronize()
        return out, {k: v.detach().clone() for k, v in m.state_dict().items()}

    (ra, sa), (rb, sb) = run(True), run(False)
    assert [round(a[1], 9) for a in ra] == [round(b[1], 9) for b in rb]            # lr after each epoch
    for a, b in zip(ra, rb):
        assert abs(a[0] - b[0]) <= 1e-6 * abs(b[0]), (a, b)                        # mean loss of the epoch
    bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
    assert not bad, bad[:6]
