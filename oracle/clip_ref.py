"""TEST INFRASTRUCTURE — CPU fp32 oracle for the CLIP ViT image/text encoders and the CLIPSeg decoder.

Functional restatement (PyTorch CPU, fp32, batch-first [B, L, D]) of:
  * CLIP.encode_text with the Long-CLIP dual positional embedding        clip/model.py:487-501, :428-431, :462-468
  * CLIPDenseBase.visual_forward with correlative self-attention (CSA) in every block and bicubic-resized
    positional embedding                                                  models/clipseg.py:79-133, :181-256
  * CLIPDensePredT.forward decoder (reduce + FiLM + post-norm TransformerEncoderLayer x3 + ConvTranspose2d)
                                                                          models/clipseg.py:436-496
(paths relative to /root/reference/).  States are flat dicts with the reference's state_dict key names.
The reference ships no weights (weights/readme.txt is a share link), so parity runs on SEEDED SYNTHETIC weights built by
make_clip_state / make_decoder_state below — the same builders run in tools/make_golden_clip.py (where the reference
consumes them through its own loader) and in the tests, so only inputs/outputs need to be stored as fixtures.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import math
import zlib

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------------
# seeded synthetic weights
# ------------------------------------------------------------------------------------------------------------
def _t(key, shape, std, seed, mean=0.0):
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7fffffff)
    return mean + std * torch.randn(*shape, generator=g)


def _fp16_round(t):
    return t.half().float()


def _resblock(s, p, width, layers, seed):
    attn_std, proj_std, fc_std = width ** -0.5, (width ** -0.5) * ((2 * layers) ** -0.5), (2 * width) ** -0.5
    # nn.MultiheadAttention / nn.Linear parameters pass through fp16 in the reference's loader (clip/model.py:631-652,689)
    s[p + ".attn.in_proj_weight"] = _fp16_round(_t(p + "ipw", (3 * width, width), attn_std, seed))
    s[p + ".attn.in_proj_bias"] = _fp16_round(_t(p + "ipb", (3 * width,), 0.02, seed))
    s[p + ".attn.out_proj.weight"] = _fp16_round(_t(p + "opw", (width, width), proj_std, seed))
    s[p + ".attn.out_proj.bias"] = _fp16_round(_t(p + "opb", (width,), 0.02, seed))
    s[p + ".ln_1.weight"] = _t(p + "l1w", (width,), 0.05, seed, 1.0)
    s[p + ".ln_1.bias"] = _t(p + "l1b", (width,), 0.02, seed)
    s[p + ".mlp.c_fc.weight"] = _fp16_round(_t(p + "fcw", (4 * width, width), fc_std, seed))
    s[p + ".mlp.c_fc.bias"] = _fp16_round(_t(p + "fcb", (4 * width,), 0.02, seed))
    s[p + ".mlp.c_proj.weight"] = _fp16_round(_t(p + "pjw", (width, 4 * width), proj_std, seed))
    s[p + ".mlp.c_proj.bias"] = _fp16_round(_t(p + "pjb", (width,), 0.02, seed))
    s[p + ".ln_2.weight"] = _t(p + "l2w", (width,), 0.05, seed, 1.0)
    s[p + ".ln_2.bias"] = _t(p + "l2b", (width,), 0.02, seed)


def make_clip_state(seed=0, vision_width=768, vision_layers=12, patch=16, grid=14, embed_dim=512, ctx=248, vocab=49408,
                    text_width=512, text_layers=12):
    """State dict of the reference's CLIP (ViT variant, Long-CLIP text positions), AFTER its loader's fp16 round trip."""
    s = {}
    vw = vision_width
    s["visual.class_embedding"] = _t("v.cls", (vw,), vw ** -0.5, seed)
    s["visual.positional_embedding"] = _t("v.pos", (grid * grid + 1, vw), vw ** -0.5, seed)
    s["visual.proj"] = _fp16_round(_t("v.proj", (vw, embed_dim), vw ** -0.5, seed))
    s["visual.conv1.weight"] = _fp16_round(_t("v.conv1", (vw, 3, patch, patch), 0.03, seed))
    for n in ("ln_pre", "ln_post"):
        s[f"visual.{n}.weight"] = _t("v." + n + "w", (vw,), 0.05, seed, 1.0)
        s[f"visual.{n}.bias"] = _t("v." + n + "b", (vw,), 0.02, seed)
    for i in range(vision_layers):
        _resblock(s, f"visual.transformer.resblocks.{i}", vw, vision_layers, seed)
    for i in range(text_layers):
        _resblock(s, f"transformer.resblocks.{i}", text_width, text_layers, seed)
    s["token_embedding.weight"] = _t("tok", (vocab, text_width), 0.02, seed)
    s["positional_embedding"] = _t("pos", (ctx, text_width), 0.01, seed)
    s["positional_embedding_res"] = _t("posres", (ctx, text_width), 0.01, seed)
    s["ln_final.weight"] = _t("lnfw", (text_width,), 0.05, seed, 1.0)
    s["ln_final.bias"] = _t("lnfb", (text_width,), 0.02, seed)
    s["text_projection"] = _fp16_round(_t("tproj", (text_width, embed_dim), text_width ** -0.5, seed))
    s["logit_scale"] = torch.tensor(math.log(1 / 0.07))
    return s


def make_decoder_state(seed=0, reduce_dim=64, depth=3, vision_width=768, ff=2048, patch=16):
    """Trainable CLIPSeg decoder parameters (models/clipseg.py:163-167, 411-418) with the reference's key names."""
    s, rd = {}, reduce_dim
    for n in ("film_mul", "film_add"):
        s[n + ".weight"] = _t(n + "w", (rd, 512), 512 ** -0.5, seed)
        s[n + ".bias"] = _t(n + "b", (rd,), 0.05, seed, 1.0 if n == "film_mul" else 0.0)
    s["reduce.weight"] = _t("redw", (rd, vision_width), vision_width ** -0.5, seed)
    s["reduce.bias"] = _t("redb", (rd,), 0.02, seed)
    for i in range(depth):
        s[f"reduces.{i}.weight"] = _t(f"r{i}w", (rd, vision_width), vision_width ** -0.5, seed)
        s[f"reduces.{i}.bias"] = _t(f"r{i}b", (rd,), 0.02, seed)
        p = f"blocks.{i}"
        s[p + ".self_attn.in_proj_weight"] = _t(p + "ipw", (3 * rd, rd), rd ** -0.5, seed)
        s[p + ".self_attn.in_proj_bias"] = _t(p + "ipb", (3 * rd,), 0.02, seed)
        s[p + ".self_attn.out_proj.weight"] = _t(p + "opw", (rd, rd), rd ** -0.5, seed)
        s[p + ".self_attn.out_proj.bias"] = _t(p + "opb", (rd,), 0.02, seed)
        s[p + ".linear1.weight"] = _t(p + "l1w", (ff, rd), rd ** -0.5, seed)
        s[p + ".linear1.bias"] = _t(p + "l1b", (ff,), 0.02, seed)
        s[p + ".linear2.weight"] = _t(p + "l2w", (rd, ff), ff ** -0.5, seed)
        s[p + ".linear2.bias"] = _t(p + "l2b", (rd,), 0.02, seed)
        for n in ("norm1", "norm2"):
            s[f"{p}.{n}.weight"] = _t(p + n + "w", (rd,), 0.05, seed, 1.0)
            s[f"{p}.{n}.bias"] = _t(p + n + "b", (rd,), 0.02, seed)
    s["trans_conv.weight"] = _t("tcw", (rd, 1, patch, patch), rd ** -0.5, seed)
    s["trans_conv.bias"] = _t("tcb", (1,), 0.02, seed)
    return s


# ------------------------------------------------------------------------------------------------------------
# functional forward
# ------------------------------------------------------------------------------------------------------------
def _ln(s, p, x):
    return F.layer_norm(x, (x.shape[-1],), s[p + ".weight"], s[p + ".bias"], 1e-5)


def _heads(t, n_heads):                      # [B, L, D] -> [B, H, L, d]
    B, L, D = t.shape
    return t.view(B, L, n_heads, D // n_heads).transpose(1, 2)


def _mlp(s, p, x):
    h = F.linear(x, s[p + ".mlp.c_fc.weight"], s[p + ".mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)                                    # QuickGELU (clip/model.py:168-170)
    return F.linear(h, s[p + ".mlp.c_proj.weight"], s[p + ".mlp.c_proj.bias"])


def csa_block(s, p, x, n_heads):
    """Residual block with correlative self-attention: softmax(q q^T s) + softmax(k k^T s) (models/clipseg.py:79-133)."""
    B, L, D = x.shape
    q, k, v = F.linear(_ln(s, p + ".ln_1", x), s[p + ".attn.in_proj_weight"], s[p + ".attn.in_proj_bias"]).chunk(3, dim=-1)
    q, k, v = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    scale = (D // n_heads) ** -0.5
    w = torch.softmax(q @ q.transpose(-1, -2) * scale, dim=-1) + torch.softmax(k @ k.transpose(-1, -2) * scale, dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, L, D)
    x = x + F.linear(o, s[p + ".attn.out_proj.weight"], s[p + ".attn.out_proj.bias"])
    return x + _mlp(s, p, _ln(s, p + ".ln_2", x))


def causal_block(s, p, x, n_heads):
    """ResidualAttentionBlock with nn.MultiheadAttention and the causal mask (clip/model.py:173-195, :462-468)."""
    B, L, D = x.shape
    q, k, v = F.linear(_ln(s, p + ".ln_1", x), s[p + ".attn.in_proj_weight"], s[p + ".attn.in_proj_bias"]).chunk(3, dim=-1)
    q, k, v = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    a = q @ k.transpose(-1, -2) * (D // n_heads) ** -0.5
    a = a + torch.full((L, L), float("-inf")).triu(1)
    o = (torch.softmax(a, dim=-1) @ v).transpose(1, 2).reshape(B, L, D)
    x = x + F.linear(o, s[p + ".attn.out_proj.weight"], s[p + ".attn.out_proj.bias"])
    return x + _mlp(s, p, _ln(s, p + ".ln_2", x))


def encode_text(s, tokens):
    """tokens int [n, 248] -> [n, embed_dim]"""
    n_layers = len({k.split(".")[2] for k in s if k.startswith("transformer.resblocks.")})
    width = s["ln_final.weight"].shape[0]
    L = tokens.shape[1]
    pos = s["positional_embedding"][:L].clone()
    pos[20:] = s["positional_embedding_res"][20:L]                        # mask1 / mask2 split at token 20 (:428-431, :490)
    x = s["token_embedding.weight"][tokens.long()] + pos
    for i in range(n_layers):
        x = causal_block(s, f"transformer.resblocks.{i}", x, width // 64)
    x = _ln(s, "ln_final", x)
    return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ s["text_projection"]


def visual_forward(s, img, extract_layers=()):
    """-> (visual_q [B, embed], activations [list of [B, L, D]])   (models/clipseg.py:188-256; activations stored batch-first)"""
    p = "visual"
    width = s[p + ".conv1.weight"].shape[0]
    n_layers = len({k.split(".")[3] for k in s if k.startswith(p + ".transformer.resblocks.")})
    x = F.conv2d(img, s[p + ".conv1.weight"], stride=s[p + ".conv1.weight"].shape[-1])
    B, _, gh, gw = x.shape
    x = x.reshape(B, width, -1).permute(0, 2, 1)
    x = torch.cat([s[p + ".class_embedding"].expand(B, 1, width), x], dim=1)
    pos = s[p + ".positional_embedding"]
    g0 = int(math.isqrt(pos.shape[0] - 1))
    if x.shape[1] != pos.shape[0]:
        grid = pos[1:].T.reshape(1, width, g0, g0)
        grid = F.interpolate(grid, (gh, gw), mode="bicubic", align_corners=False).squeeze(0).reshape(width, gh * gw).T
        pos = torch.cat([pos[:1], grid])
    x = _ln(s, p + ".ln_pre", x + pos)
    acts = []
    for i in range(n_layers):
        x = csa_block(s, f"{p}.transformer.resblocks.{i}", x, width // 64)
        if i in extract_layers:
            acts.append(x)
    q = _ln(s, p + ".ln_post", x[:, 0]) @ s[p + ".proj"]
    return q, acts


def _encoder_layer(s, p, x, n_heads):
    """nn.TransformerEncoderLayer defaults: post-norm, ReLU, eval mode (no dropout)."""
    B, L, D = x.shape
    q, k, v = F.linear(x, s[p + ".self_attn.in_proj_weight"], s[p + ".self_attn.in_proj_bias"]).chunk(3, dim=-1)
    q, k, v = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    a = torch.softmax(q @ k.transpose(-1, -2) * (D // n_heads) ** -0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, L, D)
    x = _ln(s, p + ".norm1", x + F.linear(o, s[p + ".self_attn.out_proj.weight"], s[p + ".self_attn.out_proj.bias"]))
    h = F.linear(F.relu(F.linear(x, s[p + ".linear1.weight"], s[p + ".linear1.bias"])), s[p + ".linear2.weight"], s[p + ".linear2.bias"])
    return _ln(s, p + ".norm2", x + h)


def clipseg_decoder(dec_s, acts, cond, cond_layer=0, n_heads=4):
    """The trainable part of CLIPDensePredT.forward (models/clipseg.py:452-496): acts = extracted layer activations [B, L, 768]
    in extraction order (shallow first, the forward walks them deepest first), cond [B, 512] -> [B, 1, H, W]."""
    B = acts[0].shape[0]
    a = None
    for i, act in enumerate(acts[::-1]):
        r = F.linear(act, dec_s[f"reduces.{i}.weight"], dec_s[f"reduces.{i}.bias"])
        a = r if a is None else r + a
        if i == cond_layer:
            mul = F.linear(cond, dec_s["film_mul.weight"], dec_s["film_mul.bias"])[:, None]
            add = F.linear(cond, dec_s["film_add.weight"], dec_s["film_add.bias"])[:, None]
            a = mul * a + add
        a = _encoder_layer(dec_s, f"blocks.{i}", a, n_heads)
    a = a[:, 1:].permute(0, 2, 1)
    size = int(math.isqrt(a.shape[2]))
    a = a.reshape(B, a.shape[1], size, size)
    k = dec_s["trans_conv.weight"].shape[-1]
    return F.conv_transpose2d(a, dec_s["trans_conv.weight"], dec_s["trans_conv.bias"], stride=k)


def clipseg_forward(clip_s, dec_s, img, cond, extract_layers=(3, 6, 9), cond_layer=0, n_heads=4):
    """CLIPDensePredT.forward with a [B, 512] conditional -> ([B, 1, H, W], visual_q, activations)."""
    q, acts = visual_forward(clip_s, img, extract_layers=[0] + list(extract_layers))
    return clipseg_decoder(dec_s, acts[1:], cond, cond_layer, n_heads), q, acts
