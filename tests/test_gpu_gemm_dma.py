"""The 8-wave LDS-DMA bf16 GEMM (csrc/gemm_dma.hip) behind egm_gemm: bit-identical to the register-staged kernel it replaces on the large
nn.Linear products of the CLIP ViT (clip/model.py:173-206, 487-501), and within bf16 rounding of a float64 product."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(M, N, K, act, with_bias, with_r, lda_pad=0, seed=0):
    from egm_unet_amd.clip import ops as C
    g = torch.Generator().manual_seed(seed)
    A = (torch.randn(M, K + lda_pad, generator=g) * 0.5).to(DEV).bfloat16()
    B = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    bias = torch.randn(N, generator=g).to(DEV) if with_bias else None
    R = torch.randn(M, N, generator=g).to(DEV).bfloat16() if with_r else None
    out = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV)
    C.gemm(A, K + lda_pad, B, K, True, out, N, M, N, K, torch.bfloat16, bias=bias, act=act, R=R, ldr=N)
    torch.cuda.synchronize()
    return A, B, bias, R, out


# (M, N, K): ViT-B/16 at 32 x 485 tokens (qkv / proj / fc1 / fc2: proj and fc2 take the 192-wide tiles), ragged M, a shape neither tile form
# takes, one k-chunk, many tiles per workgroup, 192-wide tiles with a ragged last row of tiles
SHAPES = [(15520, 2304, 768), (15520, 768, 768), (15520, 3072, 768), (15520, 768, 3072), (4099, 1032, 64), (8192, 2048, 128), (33000, 512, 192),
          (12000, 960, 128), (7936, 2048, 512), (7000, 2048, 128),       # + the text encoder's fc1 (248 tiles of 256 x 256: one round, not full)
          (15520, 64, 768), (15520, 128, 256), (13000, 96, 128), (7936, 512, 2048)]   # 256 x 64 / 256 x 128 tiles (reduce projections), 62 tiles of 256 x 256


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("act,with_bias,with_r", [(0, True, False), (0, True, True), (2, True, False), (1, False, True), (0, False, False)])
def test_gemm_dma_bit_identical_to_register_staged_kernel(M, N, K, act, with_bias, with_r):
    """both forms of the LDS-DMA kernel (mode 1: 8 waves, 64-row wave tiles; mode 2: 4 waves, 128-row wave tiles, accumulators in AGPRs)"""
    from egm_unet_amd._lib import lib
    L = lib()
    old = L.cdll.egm_gemm_dma_mode(-1)
    try:
        L.cdll.egm_gemm_dma_mode(0)
        A, B, bias, R, want = _run(M, N, K, act, with_bias, with_r, seed=M + N + K)
        L.cdll.egm_gemm_dma_mode(1)
        _, _, _, _, got = _run(M, N, K, act, with_bias, with_r, seed=M + N + K)
        L.cdll.egm_gemm_dma_mode(2)
        _, _, _, _, got4 = _run(M, N, K, act, with_bias, with_r, seed=M + N + K)
    finally:
        L.cdll.egm_gemm_dma_mode(old)
    assert torch.equal(got, want), f"max diff {(got.float() - want.float()).abs().max().item()}"
    assert torch.equal(got4, want), f"4-wave form: max diff {(got4.float() - want.float()).abs().max().item()}"
    # and both against float64 (sampled rows: the full product in float64 on the host is slow)
    rows = torch.linspace(0, M - 1, 97).long().to(DEV)
    ref = A[rows, :K].double() @ B.double().T
    if bias is not None:
        ref = ref + bias.double()
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    if R is not None:
        ref = ref + R[rows].double()
    err = (got[rows].double() - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) < 1.5e-2          # bf16 output rounding (2^-8 relative) + fp32 accumulation


def test_gemm_dma_strided_operands_and_fallbacks():
    """A with a row stride larger than K takes the DMA kernel too; shapes it does not support (K not a multiple of 64, few tiles, fp32 C)
    silently take the old kernels: same results with the switch on and off."""
    from egm_unet_amd._lib import lib
    L = lib()
    old = L.cdll.egm_gemm_dma_mode(-1)
    try:
        for (M, N, K, pad) in [(6000, 1024, 256, 64), (6000, 1024, 200, 0), (700, 512, 128, 0)]:
            L.cdll.egm_gemm_dma_mode(0)
            _, _, _, _, want = _run(M, N, K, 0, True, True, lda_pad=pad, seed=7)
            L.cdll.egm_gemm_dma_mode(1)
            _, _, _, _, got = _run(M, N, K, 0, True, True, lda_pad=pad, seed=7)
            assert torch.equal(got, want), (M, N, K, pad)
    finally:
        L.cdll.egm_gemm_dma_mode(old)
