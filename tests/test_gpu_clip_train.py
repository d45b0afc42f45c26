"""CLIPSeg decoder training on the HIP path (clip/train_ops.py, csrc/train_clip.hip): operator gradients against torch fp32,
the decoder's loss and parameter gradients against the fixture captured from the reference, AdamW against torch.optim.AdamW."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_layernorm_attention_film_grads(dtype):
    from egm_unet_amd.clip import train_ops as T
    g = torch.Generator().manual_seed(2)
    B, L, D, H = 3, 53, 64, 4
    tol = 2e-4 if dtype == torch.float32 else 8e-2      # bf16: ReLU masks of borderline pre-activations flip
    rd = (lambda t: t) if dtype == torch.float32 else (lambda t: t.bfloat16().float())
    x = rd(torch.randn(B, L, D, generator=g))
    Wqkv, bqkv = rd(torch.randn(3 * D, D, generator=g) / 8), torch.randn(3 * D, generator=g) * 0.1
    W1, b1 = rd(torch.randn(128, D, generator=g) / 8), torch.randn(128, generator=g) * 0.1
    W2, b2 = rd(torch.randn(D, 128, generator=g) / 11), torch.randn(D, generator=g) * 0.1
    gam, bet = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    mul, add = rd(torch.randn(B, D, generator=g)), rd(torch.randn(B, D, generator=g))
    gout = rd(torch.randn(B, L, D, generator=g))

    def run(lin, ln, att, film, xs, ps):
        x_, (Wqkv_, bqkv_, W1_, b1_, W2_, b2_, gam_, bet_, mul_, add_) = xs, ps
        a = film(x_, mul_, add_)
        o = att(lin(a, Wqkv_, bqkv_, 0, None))
        a = ln(lin(o, Wqkv_[:D], bqkv_[:D], 0, a), gam_, bet_)            # out-projection (reusing a weight slice) + residual
        h = lin(a, W1_, b1_, 1, None)
        return ln(lin(h, W2_, b2_, 0, a), gam_, bet_)

    def ref_att(qkv):
        q, k, v = [t.view(B, L, H, D // H).transpose(1, 2) for t in qkv.chunk(3, -1)]
        return (torch.softmax(q @ k.transpose(-1, -2) * (D // H) ** -0.5, -1) @ v).transpose(1, 2).reshape(B, L, D)

    ps_ref = [t.clone().requires_grad_(True) for t in (Wqkv, bqkv, W1, b1, W2, b2, gam, bet, mul, add)]
    xr = x.clone().requires_grad_(True)
    yr = run(lambda a, W, b, act, r: (F.relu(F.linear(a, W, b)) if act else F.linear(a, W, b)) + (r if r is not None else 0),
             lambda a, gm, bt: F.layer_norm(a, (D,), gm, bt, 1e-5), ref_att, lambda a, m, ad: m[:, None] * a + ad[:, None], xr, ps_ref)
    yr.backward(gout)

    def dev_param(t, act_like):
        return t.to(DEV).to(dtype if act_like else torch.float32).requires_grad_(True)
    ps_gpu = [dev_param(t, i >= 8) for i, t in enumerate((Wqkv, bqkv, W1, b1, W2, b2, gam, bet, mul, add))]
    xg = x.to(DEV).to(dtype).requires_grad_(True)
    y = run(lambda a, W, b, act, r: T.linear(a, W, b, act, r), lambda a, gm, bt: T.LayerNormFn.apply(a, gm, bt, 1e-5),
            lambda q: T.attention(q, H), lambda a, m, ad: T.FilmFn.apply(a, m, ad), xg, ps_gpu)
    assert rel(y.float(), yr.detach()) < tol
    y.backward(gout.to(DEV).to(dtype))
    assert rel(xg.grad.float(), xr.grad) < tol, "dx"
    for name, a, b in zip("Wqkv bqkv W1 b1 W2 b2 gamma beta mul add".split(), ps_gpu, ps_ref):
        assert rel(a.grad.float(), b.grad) < tol, name


def test_bce_and_transconv_grads():
    from egm_unet_amd.clip import train_ops as T
    g = torch.Generator().manual_seed(4)
    B, gsz, P, rd_ = 2, 3, 16, 64
    a = torch.randn(B, gsz * gsz + 1, rd_, generator=g)
    W, b = torch.randn(rd_, 1, P, P, generator=g) / 8, torch.randn(1, generator=g)
    tgt = (torch.rand(B, 1, gsz * P, gsz * P, generator=g) < 0.4).float()
    ar, Wr, br = a.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    grid = ar[:, 1:].permute(0, 2, 1).reshape(B, rd_, gsz, gsz)
    lr = F.binary_cross_entropy_with_logits(F.conv_transpose2d(grid, Wr, br, stride=P), tgt)
    lr.backward()
    ag, Wg, bg = a.to(DEV).requires_grad_(True), W.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = T.TransConvFn.apply(ag, Wg, bg)
    loss = T.bce_with_logits(out, tgt.to(DEV))
    loss.backward()
    assert abs(float(loss) - float(lr)) < 1e-5
    assert rel(ag.grad, ar.grad) < 1e-4 and rel(Wg.grad, Wr.grad) < 1e-4 and rel(bg.grad, br.grad) < 1e-4
    assert float(ag.grad[:, 0].abs().max()) == 0.0                     # the cls token does not reach the transposed conv


def _model(dtype):
    from oracle import clip_ref as C
    from egm_unet_amd.clipseg import CLIPDensePredT
    m = CLIPDensePredT(version="ViT-B/16", reduce_dim=64)
    m.clip_model.load_state_dict(C.make_clip_state(seed=0))
    m.load_state_dict(C.make_decoder_state(seed=0), strict=False)
    return m.to(DEV).set_compute_dtype(dtype)


def test_decoder_gradients_match_reference_fixture_fp32():
    from egm_unet_amd.clip import train_ops as T
    fx, tr = load_fixture("clipseg_fwd"), load_fixture("clipseg_train")
    m = _model(torch.float32).train()
    m.decoder_dropout = 0.0             # the fixture is the reference in eval mode with autograd on (tools/make_golden_clip.py): no dropout
    img = torch.from_numpy(fx["img"].astype(np.float32)).to(DEV)
    target = (torch.rand(2, 1, 352, 352, generator=torch.Generator().manual_seed(int(tr["target_seed"]))) < 0.3).float().to(DEV)
    out = m(img, torch.from_numpy(fx["cond"]).to(DEV))[0]
    loss = T.bce_with_logits(out, target)
    loss.backward()
    assert abs(float(loss) - float(tr["loss"])) < 5e-5, float(loss)
    params, n = dict(m.named_parameters()), 0
    for k in tr:
        if k.startswith("norm/"):
            name = k[5:]
            gflat = params[name].grad.flatten().cpu()
            ref_norm = float(tr[k])
            assert abs(float(gflat.norm()) - ref_norm) <= 5e-3 * ref_norm + 1e-7, (name, float(gflat.norm()), ref_norm)
            probe = gflat[:: max(1, gflat.numel() // 257)][:257]
            assert rel(probe, torch.from_numpy(tr["probe/" + name])) < 2e-2, name
            n += 1
    assert n == 48
    assert all(p.grad is None for k, p in params.items() if k.startswith("clip_model."))     # frozen backbone


def test_bf16_training_step_reduces_loss_and_tracks_fp32():
    from egm_unet_amd.clip import train_ops as T
    fx = load_fixture("clipseg_fwd")
    img = torch.from_numpy(fx["img"].astype(np.float32)).to(DEV)
    cond = torch.from_numpy(fx["cond"]).to(DEV)
    target = torch.zeros(2, 1, 352, 352, device=DEV); target[:, :, 100:250, 80:300] = 1.0
    losses = {}
    for dtype in (torch.float32, torch.bfloat16):
        m = _model(dtype).train()
        m.decoder_dropout = 0.0         # fp32 and bf16 runs must see the same function
        dec = [p for k, p in m.named_parameters() if p.requires_grad]
        opt = T.AdamW(dec, lr=1e-3, weight_decay=1e-2)
        ls = []
        for it in range(6):
            for gparam in opt.param_groups:
                gparam["lr"] = T.cosine_lr(1e-3, it, 6, 1e-4)
            loss = T.bce_with_logits(m(img, cond)[0], target)
            opt.zero_grad(); loss.backward(); opt.step()
            ls.append(float(loss))
        losses[dtype] = ls
        assert ls[-1] < ls[0] - 0.05, ls
    assert abs(losses[torch.bfloat16][0] - losses[torch.float32][0]) < 2e-2
    assert abs(losses[torch.bfloat16][-1] - losses[torch.float32][-1]) < 5e-2


def test_adamw_matches_torch():
    from egm_unet_amd.clip import train_ops as T
    g = torch.Generator().manual_seed(8)
    shapes = [(64, 768), (64,), (2048, 64), (64, 1, 16, 16), (1,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    o_ref = torch.optim.AdamW(ref, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    o_mine = T.AdamW(mine, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for step in range(4):
        for a, b in zip(ref, mine):
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.clone(), gr.clone().to(DEV)
        o_ref.step(); o_mine.step()
    for a, b in zip(ref, mine):
        assert torch.allclose(a.detach(), b.detach().cpu(), rtol=2e-6, atol=2e-7)
    sd = o_mine.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_kernel_statistics_and_backward_mask(dtype):
    """egm_dropout: keep rate 1 - p, survivors scaled by 1/(1-p), residual added after the mask, backward = the SAME mask on the gradient
    (regenerated from the seed), another seed = another mask, p = 0 = identity."""
    from egm_unet_amd.clip import train_ops as T
    torch.manual_seed(123)
    x = torch.ones(64, 53, 64, device=DEV, dtype=dtype, requires_grad=True)
    res = torch.full_like(x, 2.0)
    for p in (0.1, 0.5):
        y = T.dropout(x, p, residual=res)
        kept = (y.detach().float() != 2.0)
        frac = float(kept.float().mean())
        assert abs(frac - (1 - p)) < 0.01, (p, frac)
        assert torch.allclose(y.detach().float()[kept], torch.tensor(2.0 + 1.0 / (1 - p)), rtol=1e-2)
        (gx,) = torch.autograd.grad(y, x, torch.ones_like(y))
        assert torch.equal(gx.float() != 0, kept), "backward mask differs from the forward mask"
        assert torch.allclose(gx.float()[kept], torch.tensor(1.0 / (1 - p)), rtol=1e-2)
        y2 = T.dropout(x, p, residual=res)
        assert not torch.equal(y2.detach().float() != 2.0, kept), "two calls must draw different masks"
    assert torch.equal(T.dropout(x, 0.0).detach(), x.detach())


def test_decoder_dropout_train_mode():
    """train(): nn.TransformerEncoderLayer's dropout (p = 0.1) is active in the decoder like in the reference -- two forward passes
    differ, gradients are finite, p = 0 reproduces the deterministic path bit for bit; eval(): deterministic."""
    from egm_unet_amd.clip import train_ops as T
    fx = load_fixture("clipseg_fwd")
    img = torch.from_numpy(fx["img"].astype(np.float32)).to(DEV)
    cond = torch.from_numpy(fx["cond"]).to(DEV)
    target = torch.zeros(2, 1, 352, 352, device=DEV); target[:, :, 100:250, 80:300] = 1.0
    m = _model(torch.float32).train()
    assert m.decoder_dropout is None and abs(m.blocks[0].dropout.p - 0.1) < 1e-9
    torch.manual_seed(5)
    o1 = m(img, cond)[0]
    o2 = m(img, cond)[0]
    assert not torch.equal(o1, o2), "dropout must randomise the training forward"
    torch.manual_seed(5)
    o1b = m(img, cond)[0]
    assert torch.equal(o1, o1b), "same torch seed, same masks"
    loss = T.bce_with_logits(o1, target)
    loss.backward()
    grads = [p.grad for p in m.parameters() if p.requires_grad and p.grad is not None]
    assert len(grads) == 48 and all(torch.isfinite(g).all() for g in grads)     # the 48 decoder tensors the reference trains
    # the dropped model stays close to the deterministic one (p = 0.1): same sign on most pixels
    m.decoder_dropout = 0.0
    od = m(img, cond)[0]
    assert torch.equal(od, m(img, cond)[0])
    assert float(((o1 > 0) == (od > 0)).float().mean()) > 0.8
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(img, cond)[0], m(img, cond)[0])


def test_attention_dropout_gradients_match_torch_with_same_mask():
    """AttentionFn with p_drop > 0 against torch autograd on softmax(qk^T) * mask / (1-p) @ v, the mask read back from the kernel."""
    from egm_unet_amd.clip import train_ops as T
    g = torch.Generator().manual_seed(9)
    B, L, D, H, p = 2, 21, 64, 4, 0.25
    qkv = torch.randn(B, L, 3 * D, generator=g)
    gout = torch.randn(B, L, D, generator=g)
    seeds = []
    orig = T._new_seed
    T._new_seed = lambda: seeds.append(orig()) or seeds[-1]
    try:
        xq = qkv.to(DEV).requires_grad_(True)
        out = T.attention(xq, H, p)
        out.backward(gout.to(DEV))
    finally:
        T._new_seed = orig
    Lp = (L + 7) // 8 * 8
    keep = T._dropout_raw(torch.ones(B * H, L, Lp, device=DEV), p, seeds[0])[:, :, :L].cpu().view(B, H, L, L) * (1 - p)   # 0 / 1
    qr = qkv.clone().requires_grad_(True)
    q, k, v = [t.view(B, L, H, D // H).transpose(1, 2) for t in qr.chunk(3, -1)]
    P = torch.softmax(q @ k.transpose(-1, -2) * (D // H) ** -0.5, -1) * keep / (1 - p)
    (P @ v).transpose(1, 2).reshape(B, L, D).backward(gout)
    assert rel(out.detach(), (P @ v).transpose(1, 2).reshape(B, L, D).detach()) < 2e-5
    assert rel(xq.grad, qr.grad) < 1e-4


@pytest.mark.parametrize("M,N,K", [(8192, 64, 768), (9700, 2048, 64), (31040, 64, 64)])
def test_linear_weight_gradient_routes_agree(M, N, K):
    """dW = g^T x of clip/train_ops.LinearFn at decoder-training sizes by its three routes -- the conv weight-gradient kernel over the tokens
    (default), the row-split batched product, the single product -- against a float64 product of the same bf16 operands."""
    import egm_unet_amd.clip.train_ops as T
    g_ = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g_) * 0.5).to(DEV).bfloat16()
    w = (torch.randn(N, K, generator=g_) / K ** 0.5).to(DEV).requires_grad_(True)
    gy = (torch.randn(M, N, generator=g_) * 0.1).to(DEV).bfloat16()
    want = gy.double().T @ x.double()
    got = {}
    default = T._WGRAD_AS_CONV
    try:
        for route in (True, False):
            T._WGRAD_AS_CONV = route
            w.grad = None
            T.linear(x, w).backward(gy)
            torch.cuda.synchronize()
            got[route] = w.grad.double().clone()
    finally:
        T._WGRAD_AS_CONV = default
    scale = float(want.abs().max())
    for route, gw in got.items():
        assert float((gw - want).abs().max()) <= 2e-3 * scale, (route, float((gw - want).abs().max()), scale)
