"""Round-4 GPU tests: robustness of the deferred-gradient registry (ops._defer_dz) and of the gradient helpers the advisor flagged."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model_and_batch(dtype=torch.float32):
    from egm_unet_amd import GRFBUNet
    from oracle import egm_ref as R
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, 64, 64, generator=g).to(DEV)
    gl = torch.randn(2, 2, 64, 64, generator=g).to(DEV)
    m = GRFBUNet(3, 2, base_c=8)
    m.load_state_dict(R.make_egm_unet_state(3, 2, 8, seed=17), strict=True)
    return m.to(DEV).train().set_compute_dtype(dtype), x, gl


def test_aborted_backward_does_not_poison_the_deferred_dz_registry():
    """A backward that raises between a deferring producer (the classifier's data gradient, left to up4's BatchNorm backward) and its
    consumer leaves an entry in ops._DEFERRED_DZ and the engine never runs the end-of-run check.  The next backward must drop the
    stale entry, queue its own check, and give the gradients of an undisturbed model."""
    from egm_unet_amd import ops
    assert ops.fuse_dz()
    ref, x, gl = _model_and_batch()
    ref(x)["out"].backward(gl)
    want = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}

    m, _, _ = _model_and_batch()

    def boom(_g):
        raise RuntimeError("boom")
    # AccumulateGrad (and its hooks) of the classifier weight runs right after the classifier's backward, before up4's BatchNorm backward
    h = m.out_conv[0].weight.register_hook(boom)
    with pytest.raises(RuntimeError, match="boom"):
        m(x)["out"].backward(gl)
    h.remove()
    torch.cuda.synchronize()
    assert ops._DEFERRED_DZ, "the aborted run was expected to leave its deferred entry behind (otherwise this test checks nothing)"
    m.zero_grad(set_to_none=True)
    m(x)["out"].backward(gl)
    torch.cuda.synchronize()
    assert not ops._DEFERRED_DZ and ops._dz_run[0] is None
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert got.keys() == want.keys()
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_conv_bias_gradient_in_front_of_a_frozen_batchnorm_is_the_channel_sum():
    """A conv bias feeding a train-mode BatchNorm has an identically zero gradient; in front of an eval-mode (frozen) BatchNorm the true
    gradient is sum(dy) (the fused nodes used to return zeros there too)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from egm_unet_amd import ops
    from egm_unet_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(5)
    conv, bn = nn.Conv2d(8, 16, 3, padding=1, bias=True), nn.BatchNorm2d(16)
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(16, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(16, generator=g) + 0.5)
    conv.to(DEV); bn.to(DEV).eval()
    x = torch.randn(2, 12, 12, 8, generator=g).to(DEV)
    go = torch.randn(2, 12, 12, 16, generator=g).to(DEV)
    ops.conv_bn_act(x, conv, bn, ACT_RELU).backward(go)
    gb = conv.bias.grad.clone()
    conv.zero_grad(); bn.zero_grad()
    y = F.relu(bn(conv(x.permute(0, 3, 1, 2))))
    y.backward(go.permute(0, 3, 1, 2))
    assert float((gb - conv.bias.grad).norm() / conv.bias.grad.norm()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_merged_slab_launches_give_the_gradients_of_separate_launches(dtype):
    """ops.merge_wgrads: the slab kernels of the deferred weight gradients wait for the end of backward and share launches
    (egm_conv_wgrad_multi).  Same kernels on the same operands: every gradient must be bit-equal to the launch-at-once form -- which also
    shows that no x / dy tensor of a queued launch was overwritten before the launch ran."""
    from egm_unet_amd import ops
    default = ops.merge_wgrads()
    got = {}
    try:
        for merged in (False, True):
            ops.merge_wgrads(merged)
            m, x, gl = _model_and_batch(dtype)
            m(x)["out"].backward(gl)
            torch.cuda.synchronize()
            assert not ops._pending_slab_launch and not ops._pending_wgrad
            got[merged] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    finally:
        ops.merge_wgrads(default)
    assert got[True].keys() == got[False].keys() and len(got[True]) > 100
    for k in got[True]:
        assert torch.equal(got[True][k], got[False][k]), k


def test_conv_wgrad_multi_equals_single_launches_and_refuses_bad_calls():
    """egm_conv_wgrad_multi through the C ABI: four 3x3 layers of one kernel instantiation (one merged launch) and a 1x1 beside them give
    the slabs of four egm_conv_wgrad calls, bit for bit; n = 0 / n > EGM_WGRAD_MULTI_MAX are refused."""
    import struct
    from egm_unet_amd._lib import lib, ptr, stream
    L = lib()
    desc = struct.Struct("<3Q14i")
    g = torch.Generator().manual_seed(5)
    shapes = [(2, 40, 36, 32, 32, 3), (1, 64, 64, 32, 32, 3), (2, 33, 70, 32, 24, 3), (2, 16, 48, 16, 32, 3)]
    keep, blob, want = [], b"", []
    for N, H, W, Cin, Cout, K in shapes:
        x = torch.randn(N, H, W, Cin, generator=g).to(DEV).bfloat16()
        dy = torch.randn(N, H, W, Cout, generator=g).to(DEV).bfloat16()
        nfl = L.query("egm_conv_wgrad_workspace", N, H, W, Cin, Cout, K, K) // 4 + 4
        ws1 = torch.zeros(nfl, dtype=torch.float32, device=DEV); ws2 = torch.zeros_like(ws1)
        L.call("egm_conv_wgrad", 1, ptr(x), Cin, ptr(dy), Cout, None, ptr(ws1), N, H, W, Cin, Cout, Cin, Cout, K, K, 1, 1, 0, stream())
        blob += desc.pack(x.data_ptr(), dy.data_ptr(), ws2.data_ptr(), Cin, Cout, N, H, W, Cin, Cout, Cin, Cout, K, K, 1, 1, 0)
        keep.append((x, dy)); want.append((ws1, ws2))
    L.call("egm_conv_wgrad_multi", 1, blob, len(shapes), stream())
    torch.cuda.synchronize()
    for ws1, ws2 in want:
        assert torch.equal(ws1, ws2)
    for n in (0, 5):
        with pytest.raises(RuntimeError, match="conv_wgrad_multi"):
            L.call("egm_conv_wgrad_multi", 1, blob, n, stream())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_bias_gradients_equal_the_per_layer_reductions(dtype):
    """ops.defer_bgrads: the bias gradients of the convs no BatchNorm follows are summed by ONE egm_bias_grad_multi call when backward
    ends.  Same blocks, same order per tensor: every gradient of the model must be bit-equal to the per-layer form, nothing may stay
    queued, and at least one bias must really have taken the deferred path."""
    from egm_unet_amd import ops
    default = ops.defer_bgrads()
    got, queued = {}, {}
    try:
        for deferred in (False, True):
            ops.defer_bgrads(deferred)
            m, x, gl = _model_and_batch(dtype)
            seen = []
            orig = ops._flush_bgrads

            def spy(ready_only=False, _seen=seen, _orig=orig):
                _seen.append(len(ops._pending_bgrad))
                return _orig(ready_only)
            ops._flush_bgrads = spy
            try:
                m(x)["out"].backward(gl)
            finally:
                ops._flush_bgrads = orig
            torch.cuda.synchronize()
            assert not ops._pending_bgrad
            queued[deferred] = max(seen) if seen else 0
            got[deferred] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    finally:
        ops.defer_bgrads(default)
    assert queued[False] == 0 and queued[True] >= 8, queued
    assert got[True].keys() == got[False].keys() and len(got[True]) > 100
    for k in got[True]:
        assert torch.equal(got[True][k], got[False][k]), k


def test_bias_grad_multi_equals_channel_sums_and_refuses_bad_calls():
    """egm_bias_grad_multi through the C ABI against egm_channel_sums + egm_reduce_tiles per tensor (bit for bit), ragged pixel counts,
    padded channel counts, a pixel stride larger than the channel count, both dtypes."""
    import struct
    from egm_unet_amd._lib import lib, ptr, stream
    L = lib()
    ent = struct.Struct("<3Qq6i")
    g = torch.Generator().manual_seed(9)
    for code, dtype in ((0, torch.float32), (1, torch.bfloat16)):
        cases = [(2 * 33 * 17, 8, 8, 3), (4096, 64, 64, 64), (1, 16, 16, 9), (70001, 32, 48, 30), (8 * 64 * 64, 256, 256, 256)]
        keep, b1, b2, n1, n2, outs = [], b"", b"", 0, 0, []
        for npix, C, ld, Cout in cases:
            t = torch.randn(npix, ld, generator=g).to(DEV).to(dtype)
            nb = L.query("egm_channel_partials_blocks", npix, C)
            part_ref = torch.empty(nb * 2 * C, dtype=torch.float32, device=DEV)
            ref = torch.empty(2, C, dtype=torch.float32, device=DEV)
            L.call("egm_channel_sums", code, ptr(t), ld, npix, C, ptr(part_ref), stream())
            L.call("egm_reduce_tiles", ptr(part_ref), nb, C, ptr(ref), stream())
            part = torch.empty(nb * 2 * C, dtype=torch.float32, device=DEV)
            out = torch.full((C,), -7.0, dtype=torch.float32, device=DEV)
            b1 += ent.pack(t.data_ptr(), part.data_ptr(), out.data_ptr(), npix, ld, C, Cout, nb, n1, 0)
            b2 += ent.pack(t.data_ptr(), part.data_ptr(), out.data_ptr(), npix, ld, C, Cout, nb, n2, 0)
            n1 += nb; n2 += C // 8
            keep.append((t, part)); outs.append((out, ref, Cout, t, C))
        table = torch.frombuffer(bytearray(b1 + b2), dtype=torch.uint8).to(DEV)
        L.call("egm_bias_grad_multi", code, ptr(table), len(cases), n1, n2, stream())
        torch.cuda.synchronize()
        for out, ref, Cout, t, C in outs:
            assert torch.equal(out[:Cout], ref[0, :Cout])
            assert bool((out[Cout:] == -7.0).all())                       # channels beyond Cout are not written
            want = t[:, :Cout].double().sum(0)
            assert torch.allclose(out[:Cout].double(), want, rtol=1e-4, atol=1e-2 * max(1.0, float(want.abs().max())) * 1e-2)
        with pytest.raises(RuntimeError, match="bias_grad_multi"):
            L.call("egm_bias_grad_multi", code, ptr(table), 0, n1, n2, stream())
        with pytest.raises(RuntimeError, match="bias_grad_multi"):
            L.call("egm_bias_grad_multi", code, None, len(cases), n1, n2, stream())


@pytest.mark.parametrize("shape", [(2, 37, 29, 24), (1, 64, 48, 64), (2, 16, 16, 8), (1, 40, 33, 128), (1, 5, 3, 16)],
                         ids=["ragged_c24", "c64", "one_tile_c8", "c128", "tiny"])
def test_mca_backward_tiled_pass_equals_the_two_kernels(shape):
    """egm_mca_bwd_dudxo (x_out, g, codes staged with their halo in LDS, du kept there) against egm_mca_bwd_du + egm_mca_bwd_dxo on the same
    operands, bf16: same operand order in every sum, so equal up to a stray last bit from FMA contraction (as the forward's fused tail);
    ragged tiles, partial 16-channel chunks and images smaller than a tile included."""
    from egm_unet_amd._lib import lib, ptr, stream
    N, H, W, C = shape
    g_ = torch.Generator().manual_seed(sum(shape))
    xo = torch.randn(N, H, W, C, generator=g_).to(DEV).bfloat16()
    g = torch.randn(N, H, W, C, generator=g_).to(DEV).bfloat16()
    lo = torch.randint(0, 9, (N, H, W, C), generator=g_, dtype=torch.int32)
    hi = torch.randint(0, 9, (N, H, W, C), generator=g_, dtype=torch.int32)
    codes = (lo | (hi << 4)).to(torch.uint8).to(DEV)
    L, st = lib(), stream()
    du = torch.empty_like(xo); want = torch.empty_like(xo); got = torch.full_like(xo, 7.0)
    L.call("egm_mca_bwd_du", 1, ptr(xo), C, ptr(g), C, ptr(du), C, N, H, W, C, st)
    L.call("egm_mca_bwd_dxo", 1, ptr(codes), ptr(g), C, ptr(du), C, ptr(want), C, N, H, W, C, st)
    L.call("egm_mca_bwd_dudxo", 1, ptr(codes), ptr(xo), C, ptr(g), C, ptr(got), C, N, H, W, C, st)
    torch.cuda.synchronize()
    diff = (got.float() - want.float()).abs()
    assert float(diff.max()) <= 1e-2 * (1.0 + float(want.float().abs().max())), float(diff.max())
    assert float((diff > 0).float().mean()) < 0.02
    with pytest.raises(RuntimeError, match="bf16 only"):
        L.call("egm_mca_bwd_dudxo", 0, ptr(codes), ptr(xo.float()), C, ptr(g.float()), C, ptr(got.float()), C, N, H, W, C, st)


def test_mca_backward_switch_gives_the_same_model_gradients():
    """ops.fuse_mca_bwd on / off: every gradient of the bf16 model within bf16 rounding of each other."""
    from egm_unet_amd import ops
    default = ops.fuse_mca_bwd()
    got = {}
    try:
        for fused in (False, True):
            ops.fuse_mca_bwd(fused)
            m, x, gl = _model_and_batch(torch.bfloat16)
            m(x)["out"].backward(gl)
            torch.cuda.synchronize()
            got[fused] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    finally:
        ops.fuse_mca_bwd(default)
    worst = 0.0
    for k in got[True]:
        a, b = got[True][k].float(), got[False][k].float()
        worst = max(worst, float((a - b).norm() / (b.norm() + 1e-12)))
    assert worst < 0.05, worst
