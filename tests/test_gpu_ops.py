"""GPU parity tests of the primitive operators (through the C ABI) against plain PyTorch fp32 CPU references of the
same op.  fp32 path: tight tolerances; bf16 path: tolerances scaled to bf16 storage (8 significant bits)."""
import numpy as np
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from egm_unet_amd import ops
    return ops


def nhwc(x_nchw, dtype):
    """CPU NCHW fp32 -> GPU NHWC (padded to 8 channels) via torch only (independent of the kernel under test)."""
    N, C, H, W = x_nchw.shape
    CP = (C + 7) // 8 * 8
    out = torch.zeros(N, H, W, CP, dtype=dtype, device=DEV)
    out[..., :C] = x_nchw.permute(0, 2, 3, 1).to(DEV).to(dtype)
    return out


def nchw(y_nhwc, C):
    return y_nhwc[..., :C].float().permute(0, 3, 1, 2).cpu()


def tol(dtype, scale=1.0):
    return (dict(rtol=2e-4, atol=2e-4 * scale) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2 * scale))


def check(a, b, what, rtol, atol):
    a, b = a.double(), b.double()
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    assert (err <= lim).all(), (f"{what}: {int((err > lim).sum())}/{err.numel()} bad, max err {err.max():.3e} "
                                f"(allowed {lim.flatten()[err.argmax()]:.3e}), rel-L2 {((a-b).norm()/(b.norm()+1e-30)):.3e}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layout_roundtrip(dtype):
    ops = _ops()
    x = torch.randn(2, 3, 9, 13)
    y = ops.to_nhwc(x.to(DEV), dtype)
    assert y.shape == (2, 9, 13, 8)
    ref = nhwc(x, dtype)
    assert torch.equal(y.float().cpu(), ref.float().cpu())
    z = ops.to_nchw(y, 3)
    assert torch.equal(z.cpu(), ref[..., :3].float().permute(0, 3, 1, 2).cpu())


CONV_CASES = [
    # N, H, W, Cin, Cout, k, dil, groups, bias
    (2, 20, 24, 8, 16, 3, 1, 1, False),
    (2, 40, 44, 64, 64, 3, 1, 1, False),
    (1, 33, 70, 32, 40, 3, 1, 1, True),      # ragged tiles, Cout not a multiple of 32
    (2, 16, 16, 72, 24, 1, 1, 1, True),      # 1x1, Cin spans 3 chunks
    (2, 12, 14, 16, 16, 5, 1, 1, True),
    (2, 12, 14, 16, 16, 7, 1, 1, True),
    (2, 40, 44, 16, 16, 3, 12, 1, False),    # dilated (GRFB branch_dir)
    (1, 40, 44, 16, 16, 3, 36, 1, False),    # dilation ~ image size: most taps skipped
    (2, 16, 20, 8, 16, 3, 1, 8, False),      # grouped 1->2 per group (branch_edge.2)
    (2, 16, 20, 8, 16, 3, 1, 2, False),      # groups=2 (branch_ctx.1)
    (2, 18, 18, 3, 32, 3, 1, 1, False),      # Cin=3 (in_conv.0)
    (2, 18, 18, 32, 2, 1, 1, 1, True),       # OutConv
    (2, 18, 18, 64, 3, 3, 1, 1, True),       # target_enhancer
    (2, 10, 10, 2, 1, 7, 1, 1, False),       # spatial attention
    (1, 64, 64, 128, 256, 3, 1, 1, False),   # wide
    # more pixel tiles than resident workgroups: every workgroup of the persistent kernel walks several tiles
    (4, 256, 256, 32, 64, 3, 1, 1, False),   # weights staged once (single chunk), 64-cout tiles
    (3, 256, 256, 64, 64, 3, 1, 1, True),    # two chunks per tile
    (5, 256, 256, 32, 32, 1, 1, 1, True),    # 1x1
    (3, 256, 256, 16, 16, 3, 12, 1, False),  # dilated, several tiles per workgroup
    (3, 256, 256, 16, 16, 7, 1, 1, True),    # 7x7 by kernel rows
    # tall-tile (16x32, four rows per wave) variants of the 3x3 kernel: ragged edges, 1 and 2 cout blocks, 2 chunks
    (3, 250, 500, 40, 24, 3, 1, 1, True),
    (2, 250, 270, 32, 72, 3, 1, 1, True),
    (1, 512, 512, 8, 32, 3, 1, 1, False),
    # 8-wave LDS-DMA tile kernel (conv3x3_tile.hip), one case per tile shape; forward runs it, backward runs it with Cin/Cout swapped
    (2, 250, 250, 64, 128, 3, 1, 1, True),   # 16 rows x 128 couts (A), ragged right/bottom tiles, 4 chunks; dgrad: 32 rows x 64 couts
    (8, 64, 64, 32, 256, 3, 1, 1, False),    # 8 rows x 128 couts (B), two cout tiles per pixel group; dgrad: 16 rows x 32 couts
    (8, 128, 128, 32, 64, 3, 1, 1, True),    # 16 rows x 64 couts (E); dgrad: 16 rows x 32 couts (G)
    (2, 512, 512, 16, 32, 3, 1, 1, True),    # 32 rows x 32 couts (F), one chunk
    (3, 500, 260, 32, 64, 3, 1, 1, False),   # 32 rows x 64 couts (D), 432 tiles on 256 workgroups (two tiles for some), ragged
    (8, 128, 128, 48, 96, 3, 1, 1, False),   # Cout = 3 x 32: 16 rows x 32 couts with three cout tiles per group, 3 chunks
    # LDS-free activation path (conv_direct.hip): wide-in / narrow-out 1x1, ragged and 32-aligned rows; dilated on aligned rows
    (2, 40, 44, 112, 16, 1, 1, 1, True),
    # 4-row tiles of the pipelined kernel (few pixels, many channels: the bottleneck convs), ragged
    (8, 32, 32, 256, 256, 3, 1, 1, False), (2, 20, 40, 128, 96, 3, 1, 1, True),
    # more 1x1 shapes: 4-14 channel chunks, 64- and 128-cout tiles, ragged pixel counts, Cout that is no multiple of 32
    (2, 30, 30, 128, 128, 1, 1, 1, True), (1, 33, 17, 256, 192, 1, 1, 1, False), (2, 16, 16, 448, 64, 1, 1, 1, False),
    (1, 20, 20, 224, 32, 1, 1, 1, True), (3, 9, 9, 8, 8, 1, 1, 1, True), (2, 31, 31, 64, 40, 1, 1, 1, False),
    (2, 64, 64, 64, 10, 1, 1, 1, True),
    (2, 64, 64, 64, 64, 3, 12, 1, False),
    (1, 32, 32, 24, 40, 3, 24, 1, True),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_fwd_bwd(case, dtype):
    ops = _ops()
    N, H, W, Cin, Cout, k, dil, groups, bias = case
    g = torch.Generator().manual_seed(hash(case) % 2**31)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin // groups, k, k, generator=g) / (Cin // groups * k * k) ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    gy = torch.randn(N, Cout, H, W, generator=g)
    if dtype == torch.bfloat16:       # compare on bf16-representable operands so only accumulation order differs
        x, w, gy = x.bfloat16().float(), w.bfloat16().float(), gy.bfloat16().float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, padding=dil * (k - 1) // 2, dilation=dil, groups=groups)
    yr.backward(gy)

    xg = nhwc(x, dtype).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True) if bias else None
    y = ops.conv2d(xg, wg, bg, dil=dil, groups=groups)
    t = tol(dtype)
    check(nchw(y, Cout), yr.detach(), "y", **t)
    CP = y.shape[3]
    if CP > Cout:
        assert float(y[..., Cout:].float().abs().max()) == 0.0, "padded output channels must stay zero"
    y.backward(nhwc(gy, dtype))
    check(nchw(xg.grad, Cin), xr.grad, "dx", **t)
    check(wg.grad.cpu(), wr.grad, "dw", rtol=t["rtol"], atol=t["atol"] * (N * H * W) ** 0.5)
    if bias:
        check(bg.grad.cpu(), br.grad, "db", rtol=t["rtol"], atol=t["atol"] * (N * H * W) ** 0.5)


TILE_CASES = [  # N, H, W, Cin, Cout, bias: every tile shape of conv3x3_tile.hip, including the 32-cout ones the planner does not offer
    (2, 250, 250, 64, 128, True), (8, 64, 64, 32, 256, False), (4, 256, 256, 32, 64, True), (8, 128, 128, 32, 64, False),
    (2, 512, 512, 16, 32, True), (8, 128, 128, 64, 32, False), (3, 500, 260, 32, 64, False), (8, 128, 128, 48, 96, True),
    (8, 256, 256, 32, 32, False), (3, 500, 270, 32, 32, True), (5, 208, 320, 32, 32, False),     # weights-in-registers kernel (Cin = Cout = 32): even / odd tile counts per workgroup, ragged edges
]


@pytest.mark.parametrize("case", TILE_CASES, ids=[str(c) for c in TILE_CASES])
def test_conv_tile_kernel_matches_four_wave_kernel_and_torch(case):
    """The 8-wave LDS-DMA kernels with every tile shape and the weights-in-registers form enabled (egm_conv_tile_mode(7)) against the 4-wave kernel on the same bf16
    operands (same products, fp32 accumulation in another order: equal after bf16 rounding up to isolated last-bit flips), its
    BatchNorm partial sums against sums of its own output, and the output against torch's fp32 convolution."""
    ops = _ops()
    from egm_unet_amd._lib import lib, ptr, stream
    L = lib()
    N, H, W, Cin, Cout, bias = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).bfloat16().float()
    b = torch.randn(Cout, generator=g) if bias else None
    xg = nhwc(x, torch.bfloat16)
    wf, _ = ops._packed_weights(w.to(DEV), 1, torch.bfloat16)
    bg = b.to(DEV) if bias else None
    outs = []
    old = L.cdll.egm_conv_tile_mode(-1)
    try:
        for mode in (0, 7):
            L.cdll.egm_conv_tile_mode(mode)
            nt = L.query("egm_conv_stats_tiles", 1, N, H, W, Cin, Cout, 3, 3, 1)
            y = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=DEV)
            st = torch.zeros(nt, 2, Cout, dtype=torch.float32, device=DEV)
            buf = ctypes.create_string_buffer(96)
            L.cdll.egm_conv_kernel_name(1, N, H, W, Cin, Cout, 3, 3, 1, ctypes.cast(buf, ctypes.c_void_p), 96)
            L.call("egm_conv_fwd", 1, ptr(xg), Cin, ptr(wf), ptr(bg), Cout if bias else 0, ptr(y), Cout, ptr(st), N, H, W, Cin, Cout, 3, 3, 1, stream())
            torch.cuda.synchronize()
            outs.append((buf.value.decode(), y, st.sum(0)))
    finally:
        L.cdll.egm_conv_tile_mode(old)
    assert "conv3x3_" in outs[1][0] and "conv3x3_" not in outs[0][0], (outs[0][0], outs[1][0])
    y0, y1 = outs[0][1].float(), outs[1][1].float()
    assert torch.isfinite(y1).all()
    ndiff = int((y0 != y1).sum())
    assert ndiff <= y0.numel() * 1e-3 and float((y0 - y1).abs().max()) <= 2 ** -6 * float(y0.abs().max()), (ndiff, float((y0 - y1).abs().max()))
    # statistics rows = sums of the stored (bf16-rounded) output
    ref_s = torch.stack([y1.double().sum((0, 1, 2)), (y1.double() ** 2).sum((0, 1, 2))])
    assert float((outs[1][2].double() - ref_s).abs().max() / ref_s.abs().max()) < 1e-5
    yr = F.conv2d(x, w, b, padding=1)
    check(nchw(outs[1][1], Cout), yr, "y", **tol(torch.bfloat16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["relu", "sigmoid", "none"])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("C", [16, 4])
def test_bn_act(dtype, act, train, C):
    ops = _ops()
    from egm_unet_amd._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID
    code = {"relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "none": ACT_NONE}[act]
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3, C, 10, 12, generator=g) * 2 + 0.5)
    gz = torch.randn(3, C, 10, 12, generator=g)
    if dtype == torch.bfloat16:
        x, gz = x.bfloat16().float(), gz.bfloat16().float()
    bn_ref = torch.nn.BatchNorm2d(C, momentum=0.01)
    with torch.no_grad():
        bn_ref.weight.copy_(1 + 0.2 * torch.randn(C, generator=g)); bn_ref.bias.copy_(0.2 * torch.randn(C, generator=g))
        bn_ref.running_mean.copy_(0.3 * torch.randn(C, generator=g)); bn_ref.running_var.copy_(1 + 0.3 * torch.rand(C, generator=g))
    bn_gpu = torch.nn.BatchNorm2d(C, momentum=0.01)
    bn_gpu.load_state_dict(bn_ref.state_dict())
    bn_gpu.to(DEV)
    bn_ref.train(train); bn_gpu.train(train)
    fact = {"relu": torch.relu, "sigmoid": torch.sigmoid, "none": lambda v: v}[act]
    xr = x.clone().requires_grad_(True)
    zr = fact(bn_ref(xr)); zr.backward(gz)
    xg = nhwc(x, dtype).requires_grad_(True)
    z = ops.bn_act(xg, bn_gpu, code)
    z.backward(nhwc(gz, dtype))
    t = tol(dtype, 2.0)
    check(nchw(z, C), zr.detach(), "z", **t)
    check(nchw(xg.grad, C), xr.grad, "dx", **t)
    check(bn_gpu.weight.grad.cpu(), bn_ref.weight.grad, "dgamma", rtol=t["rtol"], atol=t["atol"] * 20)
    check(bn_gpu.bias.grad.cpu(), bn_ref.bias.grad, "dbeta", rtol=t["rtol"], atol=t["atol"] * 20)
    check(bn_gpu.running_mean.cpu(), bn_ref.running_mean, "running_mean", rtol=1e-4, atol=1e-4)
    check(bn_gpu.running_var.cpu(), bn_ref.running_var, "running_var", rtol=1e-4, atol=1e-4)
    assert int(bn_gpu.num_batches_tracked) == int(bn_ref.num_batches_tracked)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 19, 37), (4, 120, 250)], ids=["small", "tall_tiles"])
def test_conv_bn_stats_fused(dtype, shape):
    """conv -> BN(train, statistics from the conv epilogue) -> ReLU, forward and backward."""
    ops = _ops()
    from egm_unet_amd._lib import ACT_RELU
    import copy
    g = torch.Generator().manual_seed(9)
    x = torch.randn(shape[0], 16, shape[1], shape[2], generator=g)
    conv = torch.nn.Conv2d(16, 24, 3, padding=1, bias=False)
    bn = torch.nn.BatchNorm2d(24)
    gz = torch.randn(shape[0], 24, shape[1], shape[2], generator=g)
    if dtype == torch.bfloat16:
        x, gz = x.bfloat16().float(), gz.bfloat16().float()
        with torch.no_grad():
            conv.weight.copy_(conv.weight.bfloat16().float())
    xr = x.clone().requires_grad_(True)
    zr = torch.relu(bn(conv(xr))); zr.backward(gz)
    conv_g, bn_g = copy.deepcopy(conv).to(DEV), torch.nn.BatchNorm2d(24).to(DEV)
    conv_g.weight.grad = None
    xg = nhwc(x, dtype).requires_grad_(True)
    z = ops.conv_bn_act(xg, conv_g, bn_g, ACT_RELU)
    z.backward(nhwc(gz, dtype))
    if dtype == torch.float32:
        t = tol(dtype, 3.0)
        check(nchw(z, 24), zr.detach(), "z", **t)
        check(bn_g.running_var.cpu(), bn.running_var, "running_var", rtol=1e-4, atol=1e-4)
        check(nchw(xg.grad, 16), xr.grad, "dx", rtol=t["rtol"], atol=t["atol"] * 3)
        check(conv_g.weight.grad.cpu(), conv.weight.grad, "dw", rtol=t["rtol"], atol=t["atol"] * 40)
    else:   # bf16 storage: ReLU masks of borderline elements may flip, so compare in the aggregate
        assert rel_l2(nchw(z, 24), zr.detach()) < 1e-2
        assert rel_l2(bn_g.running_var.cpu(), bn.running_var) < 1e-2
        assert rel_l2(nchw(xg.grad, 16), xr.grad) < 4e-2
        assert rel_l2(conv_g.weight.grad.cpu(), conv.weight.grad) < 4e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 12, 20), (1, 8, 7, 9)])
def test_maxpool2(dtype, shape):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = nhwc(x, dtype).requires_grad_(True)
    y = ops.maxpool2(xg)
    y.backward(nhwc(gy, dtype))
    check(nchw(y, shape[1]), yr.detach(), "y", rtol=0, atol=0)
    check(nchw(xg.grad, shape[1]), xr.grad.bfloat16().float() if dtype == torch.bfloat16 else xr.grad, "dx", rtol=0, atol=0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shapes", [((2, 16, 10, 12), (2, 16, 20, 24)), ((1, 8, 7, 9), (1, 16, 15, 19)),
                                    # LDS-tiled backward gather (low map >= 8 x 16): ragged tiles, two channel chunks (32 + 8), padded frame
                                    ((2, 40, 20, 37), (2, 24, 41, 75)), ((1, 64, 32, 32), (1, 32, 64, 64))])
def test_upcat(dtype, shapes):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    low, skip = torch.randn(*shapes[0], generator=g), torch.randn(*shapes[1], generator=g)
    if dtype == torch.bfloat16:
        low, skip = low.bfloat16().float(), skip.bfloat16().float()
    lr_, sr_ = low.clone().requires_grad_(True), skip.clone().requires_grad_(True)
    up = F.interpolate(lr_, scale_factor=2, mode="bilinear", align_corners=True)
    dy, dx = sr_.shape[2] - up.shape[2], sr_.shape[3] - up.shape[3]
    ref = torch.cat([sr_, F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])], 1)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    lg, sg = nhwc(low, dtype).requires_grad_(True), nhwc(skip, dtype).requires_grad_(True)
    out = ops.upcat(sg, lg)
    out.backward(nhwc(go, dtype))
    t = tol(dtype)
    check(nchw(out, ref.shape[1]), ref.detach(), "out", **t)
    check(nchw(lg.grad, low.shape[1]), lr_.grad, "dlow", rtol=t["rtol"], atol=t["atol"] * 2)
    check(nchw(sg.grad, skip.shape[1]), sr_.grad, "dskip", **t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 37, 29, 24), (1, 64, 48, 64), (2, 16, 16, 8)], ids=["ragged_c24", "c64", "one_tile_c8"])
@pytest.mark.parametrize("ns", [0, 1], ids=["three_gates", "no_spatial"])
def test_mca_fused_tail_equals_the_three_kernels(dtype, shape, ns):
    """egm_mca_fused_fwd against egm_mca_xout + egm_mca_stencil1 + egm_add_avg3 on the same x and gates: x_out and the arg codes
    bit-exact, out equal up to the last bit of the storage type (same rounding points; only FMA contraction may differ), including
    ragged tiles (H, W not multiples of 16) and a partial 32-channel chunk.  Also the in-place upsample path against the copy path."""
    from egm_unet_amd._lib import dtype_code, lib, ptr, stream
    N, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, H, W, C, generator=g).to(DEV).to(dtype)
    gates = torch.rand(N, H + W + C, generator=g)
    if ns:
        gates[:, H + W:] = 0                                  # what egm_mca_gates_fwd writes for ks_c == 0
    gates = gates.to(DEV)
    L, dt, st = lib(), dtype_code(dtype), stream()
    xo_a, r1, u2, out_a = (torch.empty_like(x) for _ in range(4))
    codes_a = torch.empty((N, H, W, C), dtype=torch.uint8, device=DEV)
    L.call("egm_mca_xout", dt, ptr(x), C, ptr(gates), ptr(xo_a), C, N, H, W, C, ns, st)
    L.call("egm_mca_stencil1", dt, ptr(xo_a), C, ptr(r1), C, ptr(u2), C, ptr(codes_a), N, H, W, C, st)
    L.call("egm_add_avg3", dt, ptr(r1), C, ptr(u2), C, 0.2, ptr(out_a), C, N, H, W, C, st)
    xo_b, out_b = torch.empty_like(x), torch.empty_like(x)
    codes_b = torch.empty_like(codes_a)
    L.call("egm_mca_fused_fwd", dt, ptr(x), C, ptr(gates), ptr(xo_b), C, ptr(out_b), C, ptr(codes_b), N, H, W, C, ns, st)
    torch.cuda.synchronize()
    assert torch.equal(xo_a, xo_b)
    assert torch.equal(codes_a, codes_b)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    diff = (out_a.float() - out_b.float()).abs()
    assert float(diff.max()) <= tol * (1.0 + float(out_a.float().abs().max())), float(diff.max())
    assert float((diff > 0).float().mean()) < 0.02          # at most a stray last-bit difference here and there
    # x_out may be dropped (inference)
    out_c = torch.empty_like(x)
    L.call("egm_mca_fused_fwd", dt, ptr(x), C, ptr(gates), None, C, ptr(out_c), C, None, N, H, W, C, ns, st)
    torch.cuda.synchronize()
    assert torch.equal(out_b, out_c)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upcat_in_place_equals_copy(dtype):
    """upcat into a concat buffer that already holds the skip (cat_slots) == upcat that copies the skip; gradients too."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    skip = torch.randn(2, 20, 24, 16, generator=g).to(DEV).to(dtype)
    low = torch.randn(2, 10, 12, 8, generator=g).to(DEV).to(dtype)
    s1, l1 = skip.clone().requires_grad_(True), low.clone().requires_grad_(True)
    ref = ops.upcat(s1, l1)
    buf, (slot, _) = ops.cat_slots(2, 20, 24, [16, 8], dtype, DEV)
    l2 = low.clone().requires_grad_(True)
    ops._axpby(skip, 1.0, None, 0.0, slot)                  # stands in for a producer that writes with out=slot
    got = ops.upcat(slot, l2, buf)
    assert got.data_ptr() == buf.data_ptr()
    assert torch.equal(ref.detach(), got.detach())
    gout = torch.randn(ref.shape, generator=g).to(DEV).to(dtype)
    ref.backward(gout)
    got.backward(gout)
    assert torch.equal(l1.grad, l2.grad)
