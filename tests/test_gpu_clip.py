"""GPU parity of the CLIP / CLIPSeg inference path: transformer kernels against plain PyTorch fp32 references, encoders and
the full CLIPSeg forward against fixtures captured from the reference (seeded synthetic weights)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close, load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(300, 96, 64, True), (485, 2304, 768, True), (77, 64, 485, False), (130, 256, 64, False), (33, 40, 24, True)])
def test_gemm_bias_act_residual(dtype, shape):
    from egm_unet_amd.clip import ops as O
    M, N, K, transB = shape
    g = torch.Generator().manual_seed(M * 7 + N)
    Kp = (K + 7) // 8 * 8
    A = torch.zeros(M, Kp); A[:, :K] = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g) / K ** 0.5 if transB else torch.randn(K, N, generator=g) / K ** 0.5
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    if dtype == torch.bfloat16:
        A, Bm, R = A.bfloat16().float(), Bm.bfloat16().float(), R.bfloat16().float()
    ref = A[:, :K] @ (Bm.T if transB else Bm) * 0.5 + bias
    ref = ref * torch.sigmoid(1.702 * ref) + R
    Ad, Rd = A.to(DEV).to(dtype), R.to(DEV).to(dtype)
    if transB:
        Bp = torch.zeros(N, Kp); Bp[:, :K] = Bm
        Bd, ldb = Bp.to(DEV).to(dtype), Kp
    else:
        Bd, ldb = Bm.to(DEV).to(dtype), N
    C = torch.empty(M, N, dtype=dtype, device=DEV)
    O.gemm(Ad, Kp, Bd, ldb, transB, C, N, M, N, K, dtype, bias=bias.to(DEV), act=2, R=Rd, ldr=N, alpha=0.5)
    assert rel(C.float(), ref) < (2e-5 if dtype == torch.float32 else 6e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_modes_vs_torch(dtype):
    from egm_unet_amd.clip import ops as O
    g = torch.Generator().manual_seed(3)
    B, L, H, dh = 2, 101, 4, 16
    D = H * dh
    qkv = torch.randn(B, L, 3 * D, generator=g)
    if dtype == torch.bfloat16:
        qkv = qkv.bfloat16().float()
    q, k, v = [t.view(B, L, H, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    s = dh ** -0.5
    refs = {
        "full": torch.softmax(q @ k.transpose(-1, -2) * s, -1) @ v,
        "causal": torch.softmax(q @ k.transpose(-1, -2) * s + torch.full((L, L), float("-inf")).triu(1), -1) @ v,
        "csa": (torch.softmax(q @ q.transpose(-1, -2) * s, -1) + torch.softmax(k @ k.transpose(-1, -2) * s, -1)) @ v,
    }
    for mode, ref in refs.items():
        out = O.attention(qkv.to(DEV).to(dtype), H, mode)
        assert rel(out.float(), ref.transpose(1, 2).reshape(B, L, D)) < (2e-5 if dtype == torch.float32 else 1e-2), mode


@pytest.mark.parametrize("L", [485, 248, 101, 33])
def test_fused_attention_head64_vs_torch(L):
    """bf16, head dimension 64 (ViT-B/16 and the text encoder): the fused kernel (scores on chip) against fp32 torch."""
    from egm_unet_amd.clip import ops as O
    g = torch.Generator().manual_seed(L)
    B, H, dh = 2, 3, 64
    D = H * dh
    qkv = (torch.randn(B, L, 3 * D, generator=g) * 1.5).bfloat16().float()
    q, k, v = [t.view(B, L, H, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    s = dh ** -0.5
    refs = {
        "full": torch.softmax(q @ k.transpose(-1, -2) * s, -1) @ v,
        "causal": torch.softmax(q @ k.transpose(-1, -2) * s + torch.full((L, L), float("-inf")).triu(1), -1) @ v,
        "csa": (torch.softmax(q @ q.transpose(-1, -2) * s, -1) + torch.softmax(k @ k.transpose(-1, -2) * s, -1)) @ v,
    }
    for mode, ref in refs.items():
        out = O.attention(qkv.to(DEV).bfloat16(), H, mode)
        assert out.shape == (B, L, D)
        assert rel(out.float(), ref.transpose(1, 2).reshape(B, L, D)) < 1e-2, (mode, L)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [768, 512, 64, 1024, 2048, 100, 4096])    # register-resident vector path (ragged lane counts, its limit) and the scalar loop
def test_layernorm_vs_torch(dtype, D):
    from egm_unet_amd.clip import ops as O
    g = torch.Generator().manual_seed(4)
    x = torch.randn(37, D, generator=g) * 3 + 1
    ln = torch.nn.LayerNorm(D)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(D, generator=g)); ln.bias.copy_(0.1 * torch.randn(D, generator=g))
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    y = O.layernorm(x.to(DEV).to(dtype), ln.to(DEV))
    assert rel(y.float(), ln.cpu()(x).detach()) < (1e-5 if dtype == torch.float32 else 5e-3)


@pytest.fixture(scope="module")
def clipseg():
    from oracle import clip_ref as C
    from egm_unet_amd.clipseg import CLIPDensePredT
    m = CLIPDensePredT(version="ViT-B/16", reduce_dim=64)
    m.clip_model.load_state_dict(C.make_clip_state(seed=0))
    res = m.load_state_dict(C.make_decoder_state(seed=0), strict=False)
    assert not res.unexpected_keys
    return m.to(DEV).eval()


def test_text_encoder_fixture(clipseg):
    fx, tk = load_fixture("clipseg_fwd"), load_fixture("clip_tokens")
    clipseg.set_compute_dtype(torch.float32)
    feats = clipseg.clip_model.encode_text(torch.from_numpy(tk["tokens248"][:6]))
    assert_close(feats.cpu(), fx["text_feats"], rtol=1e-3, atol=1e-4, what="text features (fp32)")
    clipseg.set_compute_dtype(torch.bfloat16)
    f16 = clipseg.clip_model.encode_text(torch.from_numpy(tk["tokens248"][:6]))
    assert rel(f16, torch.from_numpy(fx["text_feats"])) < 3e-2
    clipseg.set_compute_dtype(torch.float32)


def test_clipseg_forward_fixture_fp32(clipseg):
    fx = load_fixture("clipseg_fwd")
    prompts = open(__import__("os").path.join(__import__("helpers").GOLDEN, "clip_prompts.txt"), encoding="utf-8").read().split("\n")
    img = torch.from_numpy(fx["img"].astype(np.float32)).to(DEV)
    clipseg.set_compute_dtype(torch.float32)
    out, q, cond, acts = clipseg(img, [prompts[0], prompts[3]], return_features=True)
    assert_close(cond.cpu(), fx["cond"], rtol=1e-3, atol=1e-4, what="cond")
    assert_close(q.cpu(), fx["visual_q"], rtol=1e-3, atol=1e-3, what="visual_q")
    for i, a in enumerate(acts):                       # [L, B, D] like the reference
        assert_close(a[0].cpu(), fx[f"act{i}_cls"], rtol=2e-3, atol=2e-3, what=f"act{i} cls")
        assert_close(a[1:9].permute(1, 0, 2).cpu(), fx[f"act{i}_tok"], rtol=2e-3, atol=2e-3, what=f"act{i} tokens")
    assert out.shape == (2, 1, 352, 352)
    assert_close(out[:, :, ::4, ::4].cpu(), fx["out"], rtol=1e-3, atol=2e-3, what="mask logits (subsampled)")
    assert_close(out[:, :, 100:164, 100:164].cpu(), fx["out_crop"], rtol=1e-3, atol=2e-3, what="mask logits (crop)")
    sign_agree = float(((out[:, :, ::4, ::4].cpu() > 0) == (torch.from_numpy(fx["out"]) > 0)).float().mean())
    assert sign_agree > 0.999, sign_agree
    # 224x224: 197 tokens, stored positional embedding (no resize); conditional passed as a tensor
    o224 = clipseg(torch.from_numpy(fx["img224"].astype(np.float32)).to(DEV), torch.from_numpy(fx["cond"][:1]).to(DEV))[0]
    assert_close(o224[:, :, 64:128, 64:128].cpu(), fx["out224_crop"], rtol=1e-3, atol=2e-3, what="224 crop")


def test_clipseg_forward_bf16_tracks_fp32(clipseg):
    fx = load_fixture("clipseg_fwd")
    img = torch.from_numpy(fx["img"].astype(np.float32)).to(DEV)
    clipseg.set_compute_dtype(torch.bfloat16)
    out = clipseg(img, torch.from_numpy(fx["cond"]).to(DEV))[0]
    clipseg.set_compute_dtype(torch.float32)
    ref = torch.from_numpy(fx["out"])
    assert rel(out[:, :, ::4, ::4], ref) < 0.1
    assert float(((out[:, :, ::4, ::4].cpu() > 0) == (ref > 0)).float().mean()) > 0.95


def test_ensemble_fuse_and_alpha_search_vs_torch():
    """fused = bilinear(clip) + alpha*unet, argmax, and the alpha grid search against the same computation in torch."""
    from egm_unet_amd.ensemble import fuse_predict, search_best_alpha
    g = torch.Generator().manual_seed(11)
    clips = [torch.randn(1, 2, 352, 352, generator=g) for _ in range(2)]
    unets = [torch.randn(1, 2, 200, 260, generator=g) * 0.3 for _ in range(2)]
    labels = [torch.randint(0, 2, (200, 260), generator=g) for _ in range(2)]
    for lab in labels:
        lab[:5] = 255
    up = [F.interpolate(c, size=(200, 260), mode="bilinear", align_corners=False) for c in clips]
    pred, fused = fuse_predict(clips[0].to(DEV), unets[0].to(DEV), 3.5, return_fused=True)
    ref = up[0] + 3.5 * unets[0]
    assert_close(fused.cpu(), ref, rtol=1e-5, atol=1e-5, what="fused logits")
    assert float((pred.cpu() == ref.argmax(1)).float().mean()) > 0.9999
    best, best_miou, mious = search_best_alpha([c.to(DEV) for c in clips], [u.to(DEV) for u in unets], [l.numpy() for l in labels],
                                               search_scale=(0.1, 10.0), search_step=100)
    alphas = np.linspace(0.1, 10.0, 100)
    ref_m = []
    for a in alphas:
        mat = torch.zeros(2, 2)
        for u_, c_, l_ in zip(unets, up, labels):
            p = (c_ + float(np.float32(a)) * u_).argmax(1).flatten(); t = l_.flatten(); k = (t >= 0) & (t < 2)
            mat += torch.bincount(2 * t[k] + p[k], minlength=4).reshape(2, 2).float()
        iu = torch.diag(mat) / (mat.sum(1) + mat.sum(0) - torch.diag(mat))
        ref_m.append(float(iu.mean()))
    assert np.allclose(mious, np.array(ref_m), atol=2e-4), float(np.abs(mious - np.array(ref_m)).max())
    assert abs(best_miou - max(ref_m)) < 2e-4


def test_ensemble_matches_reference_fixture():
    """N2 pinned by the reference itself: tests/golden/ensemble_alpha.npz holds what /root/reference/eval_CLIPseg.py's own
    search_best_alpha + ConfusionMatrix (:656-749) and its fusion lines (:884-910) computed on seeded logits
    (tools/make_golden_ensemble.py): bilinear resize of the CLIPSeg logits, per-alpha global mIoU over the validation set, the best
    alpha (first maximum), and argmax(clip + alpha * unet)."""
    from helpers import load_fixture
    from egm_unet_amd.ensemble import fuse_predict, search_best_alpha
    fx = load_fixture("ensemble_alpha")
    clips = [torch.from_numpy(c).to(DEV) for c in fx["clip"]]
    unets = [torch.from_numpy(u).to(DEV) for u in fx["unet"]]
    labels = [lab for lab in fx["labels"]]
    best, best_miou, mious = search_best_alpha(clips, unets, labels, search_scale=(0.1, 10.0), search_step=100)
    assert np.allclose(mious, fx["mious"], atol=1e-5), float(np.abs(mious - fx["mious"]).max())
    assert abs(best - float(fx["best_alpha"])) < 1e-9, (best, float(fx["best_alpha"]))
    for i in range(len(clips)):
        pred, fused = fuse_predict(clips[i], unets[i], best, return_fused=True)
        ref_fused = torch.from_numpy(fx["resized"][i]) + float(np.float32(best)) * torch.from_numpy(fx["unet"][i])
        assert_close(fused.cpu(), ref_fused, rtol=1e-5, atol=1e-5, what="fused logits")
        agree = float((pred.cpu()[0] == torch.from_numpy(fx["pred"][i].astype(np.int64))).float().mean())
        assert agree > 0.9995, agree          # argmax of logits that differ in the last fp32 bit may flip on exact near-ties only


def test_clipseg_masked_matches_reference_fixture():
    """CLIPDensePredTMasked (models/clipseg.py:500-525): conditional vector from a support image with the class token's attention masked
    by its segmentation in every layer (batch of 2: the reference pairs masks with heads through attn_mask.repeat(n_heads, 1)), then the
    query decode -- against tests/golden/clipseg_masked.npz, produced by the reference's own class (tools/make_golden_clip_masked.py)."""
    from oracle import clip_ref as C
    from egm_unet_amd.clipseg import CLIPDensePredTMasked
    fx = load_fixture("clipseg_masked")
    m = CLIPDensePredTMasked(version="ViT-B/16", reduce_dim=64)
    m.clip_model.load_state_dict(C.make_clip_state(seed=0))
    m.load_state_dict(C.make_decoder_state(seed=0), strict=False)
    m.to(DEV).eval()
    img_q = torch.from_numpy(fx["img_q"].astype(np.float32)).to(DEV)
    img_s = torch.from_numpy(fx["img_s"].astype(np.float32)).to(DEV)
    seg = torch.from_numpy(fx["seg"].astype(np.float32)).to(DEV)
    with torch.no_grad():
        cond, _, _ = m.visual_forward_masked(img_s, seg)
        out = m(img_q, img_s, seg)[0]
    assert_close(cond.cpu(), fx["cond"], rtol=2e-3, atol=2e-3, what="masked conditional")
    # the mask matters: the result is far closer to the masked fixture than to the unmasked conditional
    assert rel(cond.cpu(), torch.from_numpy(fx["cond"])) < 0.1 * rel(torch.from_numpy(fx["cond_plain"]), torch.from_numpy(fx["cond"]))
    assert_close(out[:, :, ::4, ::4].cpu(), fx["out"], rtol=2e-3, atol=3e-3, what="mask logits (subsampled)")
    assert_close(out[:, :, 100:164, 100:164].cpu(), fx["out_crop"], rtol=2e-3, atol=3e-3, what="mask logits (crop)")
    # bf16 path: the probabilities are materialised for the mask (unfused attention); it tracks fp32
    m.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        cond_bf, _, _ = m.visual_forward_masked(img_s, seg)
    assert rel(cond_bf.cpu(), torch.from_numpy(fx["cond"])) < 0.05
