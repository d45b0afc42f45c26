"""The weights-in-registers 7x7 kernel for 16 -> 16 channels (csrc/conv7x7_c16.hip; FusionConv's merged multi-scale conv at the
64-channel level, src/EGM-UNet.py:1210-1228) against the generic pipelined kernel and against torch, forward and data gradient."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("shape", [(2, 40, 100), (1, 8, 64), (1, 5, 7), (3, 64, 192), (2, 33, 129)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_conv7x7_c16_matches_generic_kernel_and_torch(shape, with_bias):
    from egm_unet_amd import ops
    from egm_unet_amd._lib import dtype_code, lib
    N, H, W = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, H, W, 16, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(16, 16, 7, 7, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b = (torch.randn(16, generator=g) * 0.1).to(DEV).requires_grad_(True) if with_bias else None
    gy = torch.randn(N, H, W, 16, generator=g).to(DEV).to(torch.bfloat16)
    L = lib()
    buf = __import__("ctypes").create_string_buffer(96)
    L.cdll.egm_conv_kernel_name(dtype_code(torch.bfloat16), N, H, W, 16, 16, 7, 7, 1, buf, 96)
    assert buf.value.decode() == "conv7x7_c16_kernel"       # the kernel under test is the one the shape takes
    res = []
    old = L.cdll.egm_conv_c7_mode(-1)
    try:
        for mode in (1, 0):
            L.cdll.egm_conv_c7_mode(mode)
            ops.bump_weight_generation()
            w.grad = None
            xa = x.clone().requires_grad_(True)
            y = ops.conv2d(xa, w, b)
            y.backward(gy)
            torch.cuda.synchronize()
            res.append((y.detach().clone(), xa.grad.clone(), w.grad.clone()))
    finally:
        L.cdll.egm_conv_c7_mode(old)
        ops.bump_weight_generation()
    (y1, dx1, dw1), (y0, dx0, dw0) = res
    # same products, fp32 accumulation in a different order: agreement to bf16 rounding of the result
    for a, r, what in ((y1, y0, "y"), (dx1, dx0, "dx")):
        err = float((a.float() - r.float()).abs().max()) / max(1e-6, float(r.float().abs().max()))
        assert err <= 1e-2, (what, err)
    # the weight gradient takes conv7x7_c16_wgrad_kernel under mode 1: same bf16 products, fp32 sums in another order
    werr = float((dw1 - dw0).abs().max()) / max(1e-12, float(dw0.abs().max()))
    assert werr <= 2e-5, werr
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float()
    yr = F.conv2d(xr, wr, None if b is None else b.detach(), padding=3)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    ref, dref = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)
    assert float((y1.float() - ref).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max()))
    assert float((dx1.float() - dref).abs().max()) <= 1e-2 * max(1.0, float(dref.abs().max()))
    wref = F.conv2d  # (autograd reference of dW: bf16 operands, fp32 accumulation)
    xw = x.float().permute(0, 3, 1, 2)
    ww = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    F.conv2d(xw, ww, None, padding=3).backward(gy.float().permute(0, 3, 1, 2))
    assert float((dw1 - ww.grad).abs().max()) <= 2e-3 * max(1e-6, float(ww.grad.abs().max()))
    del wref


@pytest.mark.parametrize("shape", [(8, 128, 128, 32, 64, 32), (2, 256, 256, 32, 64, 32), (8, 128, 128, 128, 256, 128), (8, 100, 132, 64, 128, 64), (6, 129, 131, 32, 64, 32)])
def test_conv_fwd_split_equals_one_output(shape):
    """egm_conv_fwd_split (the data gradient behind a channel concatenation written as two dense tensors) == the channel slices of
    egm_conv_fwd's single output, bit for bit."""
    from egm_unet_amd._lib import dtype_code, lib, ptr, stream
    N, H, W, Cin, Cout, cs = shape
    L, dt = lib(), dtype_code(torch.bfloat16)
    assert L.cdll.egm_conv_split_ok(dt, N, H, W, Cin, Cout, 3, 3, 1, cs), "shape chosen to take the 8-wave tile kernel"
    assert not L.cdll.egm_conv_split_ok(dt, 4, 64, 64, 128, 256, 3, 3, 1, 128)        # too few tiles for that kernel: callers fall back to one output
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(DEV)
    wf = torch.empty(9 * Cout * Cin, dtype=torch.bfloat16, device=DEV)
    L.call("egm_conv_pack", dt, ptr(w), ptr(wf), None, Cout, Cin, 3, 3, 1, stream())
    y = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    L.call("egm_conv_fwd", dt, ptr(x), Cin, ptr(wf), None, 0, ptr(y), Cout, None, N, H, W, Cin, Cout, 3, 3, 1, stream())
    ya = torch.full((N, H, W, cs), 7.0, dtype=torch.bfloat16, device=DEV)
    yb = torch.full((N, H, W, Cout - cs), 7.0, dtype=torch.bfloat16, device=DEV)
    L.call("egm_conv_fwd_split", dt, ptr(x), Cin, ptr(wf), ptr(ya), cs, ptr(yb), Cout - cs, cs, N, H, W, Cin, Cout, 3, 3, 1, stream())
    torch.cuda.synchronize()
    assert torch.equal(ya, y[..., :cs]) and torch.equal(yb, y[..., cs:])


@pytest.mark.parametrize("shape", [(2, 64, 160), (1, 40, 100), (2, 37, 129), (1, 128, 128)])
@pytest.mark.parametrize("dil", [12, 24, 36])
def test_dilated_c16_kernels_match_generic_kernels_and_torch(shape, dil):
    """conv3x3d_c16_kernel / conv3x3d_c16_wgrad_kernel (16 -> 16 channels, dilation 12 / 24 / 36: the EdgeEnhancedGRFB branch convs,
    src/EGM-UNet.py:1256-1278): forward with statistics, data gradient and weight gradient against the generic kernels and torch."""
    from egm_unet_amd import ops
    from egm_unet_amd._lib import lib
    N, H, W = shape
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, H, W, 16, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(16, 16, 3, 3, generator=g) * 0.1).to(DEV).requires_grad_(True)
    gy = torch.randn(N, H, W, 16, generator=g).to(DEV).to(torch.bfloat16)
    L = lib()
    old = L.cdll.egm_conv_c7_mode(-1)
    res = []
    try:
        for mode in (1, 0):
            L.cdll.egm_conv_c7_mode(mode)
            ops.bump_weight_generation()
            w.grad = None
            xa = x.clone().requires_grad_(True)
            y, stats = ops.conv2d(xa, w, None, dil=dil, want_stats=True)          # forward with BatchNorm partial sums, as BasicConv runs it
            y.backward(gy)
            torch.cuda.synchronize()
            res.append((w.grad.clone(), y.detach().clone(), xa.grad.clone(), stats.double().sum(0).clone()))
    finally:
        L.cdll.egm_conv_c7_mode(old)
        ops.bump_weight_generation()
    (dw1, y1, dx1, st1), (dw0, y0, dx0, st0) = res
    werr = float((dw1 - dw0).abs().max()) / max(1e-12, float(dw0.abs().max()))
    assert werr <= 2e-5, werr
    for a, r, what in ((y1, y0, "y"), (dx1, dx0, "dx")):                           # conv3x3d_c16_kernel vs the LDS-free kernel
        err = float((a.float() - r.float()).abs().max()) / max(1e-6, float(r.float().abs().max()))
        assert err <= 1e-2, (what, err)
    # the statistics rows are the sums of the ROUNDED outputs the kernel stored
    ref_st = torch.stack([y1.double().sum((0, 1, 2)), (y1.double() ** 2).sum((0, 1, 2))])
    assert float((st1 - ref_st).abs().max()) <= 1e-4 * float(ref_st.abs().max())
    ww = w.detach().clone().requires_grad_(True)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.conv2d(xr, ww, None, padding=dil, dilation=dil)
    yr.backward(gy.float().permute(0, 3, 1, 2))
    assert float((dw1 - ww.grad).abs().max()) <= 1e-4 * max(1e-6, float(ww.grad.abs().max()))
    wb = w.detach().to(torch.bfloat16).float()
    yb = F.conv2d(x.float().permute(0, 3, 1, 2), wb, None, padding=dil, dilation=dil).permute(0, 2, 3, 1)
    assert float((y1.float() - yb).abs().max()) <= 1e-2 * max(1.0, float(yb.abs().max()))
