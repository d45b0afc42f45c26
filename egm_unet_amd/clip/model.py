"""CLIP (ViT image encoder + Long-CLIP text encoder) with the reference's parameter layout (clip/model.py:173-206,
209-355, 358-501, 654-691) computed by the HIP library.  nn.* children are parameter holders named like the
reference's, so `build_model(state_dict)` accepts reference checkpoints; forward passes run batch-first token matrices
through clip/ops.py.  Inference only (the backbone is frozen in CLIPSeg)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops as O
from .._lib import dtype_code, lib, ptr, require_gpu, stream


class LayerNorm(nn.LayerNorm):
    pass


class QuickGELU(nn.Module):
    pass


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.n_head = n_head

    def run(self, x, mode, cls_mask=None):
        """x [B, L, D]; mode 'csa' | 'causal' | 'full'; cls_mask: optional [B, L-1] fp32 multiplier of the class token's attention row."""
        qkv = O.linear(O.layernorm(x, self.ln_1), self.attn.in_proj_weight, self.attn.in_proj_bias)
        a = O.attention(qkv, self.n_head, mode, cls_mask)
        x = O.linear(a, self.attn.out_proj.weight, self.attn.out_proj.bias, residual=x)
        h = O.linear(O.layernorm(x, self.ln_2), self.mlp.c_fc.weight, self.mlp.c_fc.bias, act=2)          # QuickGELU fused
        return O.linear(h, self.mlp.c_proj.weight, self.mlp.c_proj.bias, residual=x)


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, attn_mask=None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution, self.patch_size, self.output_dim, self.width, self.heads = input_resolution, patch_size, output_dim, width, heads
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._pos_cache = {}

    def positional_embedding_for(self, gh, gw):
        """Positional embedding for a gh x gw token grid: the stored one, or its bicubic resize (models/clipseg.py:181-186).
        One-time parameter preprocessing per resolution (cached), outside the per-image path."""
        pos = self.positional_embedding.detach()
        g0 = int(round((pos.shape[0] - 1) ** 0.5))
        if (gh, gw) == (g0, g0):
            return pos.float().contiguous()
        key = (gh, gw, pos.data_ptr(), pos._version)
        if key not in self._pos_cache:
            grid = pos[1:].T.reshape(1, self.width, g0, g0).float()
            grid = F.interpolate(grid, (gh, gw), mode="bicubic", align_corners=False).squeeze(0).reshape(self.width, gh * gw).T
            self._pos_cache = {key: torch.cat([pos[:1].float(), grid]).contiguous()}
        return self._pos_cache[key]

    def run(self, img, dtype, extract_layers=(), csa_all_layers=True, cls_mask=None):
        """img fp32 NCHW -> (cls feature [B, output_dim], [activations [B, L, D] at extract_layers]).
        cls_mask = (layer index | 'all', [B, tokens] fp32): see CLIPDensePredT.visual_forward."""
        B, C, H, W = img.shape
        P = self.patch_size
        gh, gw = H // P, W // P
        L_, code = lib(), dtype_code(dtype)
        patches = torch.empty((B * gh * gw, C * P * P), dtype=dtype, device=img.device)
        L_.call("egm_patchify", code, ptr(img.contiguous().float()), ptr(patches), B, C, H, W, P, stream())
        tok = O.linear(patches, self.conv1.weight.reshape(self.width, -1))
        x = torch.empty((B, gh * gw + 1, self.width), dtype=dtype, device=img.device)
        L_.call("egm_vit_assemble", code, ptr(tok), ptr(self.class_embedding.detach().float()), ptr(self.positional_embedding_for(gh, gw)), ptr(x),
                B, gh * gw, self.width, stream())
        x = O.layernorm(x, self.ln_pre)
        acts = []
        n = len(self.transformer.resblocks)
        for i, blk in enumerate(self.transformer.resblocks):
            m = cls_mask[1] if (cls_mask is not None and cls_mask[0] in ("all", i)) else None
            x = blk.run(x, "csa" if (csa_all_layers or i == n - 1) else "full", m)
            if i in extract_layers:
                acts.append(x)
        cls = O.layernorm(x[:, 0].contiguous(), self.ln_post)
        return O.matmul_kn(cls, self.proj), acts


class CLIP(nn.Module):
    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length, vocab_size,
                 transformer_width, transformer_heads, transformer_layers, load_from_clip=False):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("egm_unet_amd: only the ViT image encoder is implemented (ModifiedResNet is unused by CLIPSeg)")
        self.context_length = 248
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers, vision_width // 64, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        n_pos = 77 if load_from_clip else 248
        self.positional_embedding = nn.Parameter(torch.empty(n_pos, transformer_width))
        if not load_from_clip:
            self.positional_embedding_res = nn.Parameter(torch.empty(248, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        if not load_from_clip:
            nn.init.normal_(self.positional_embedding_res, std=0.01)
        nn.init.normal_(self.text_projection, std=transformer_width ** -0.5)
        self.compute_dtype = torch.float32

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        return self

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    @torch.no_grad()
    def encode_text(self, text):
        """text int [n, L<=248] (clip.tokenize) -> [n, embed_dim] fp32   (clip/model.py:487-501)"""
        require_gpu()
        dev = self.token_embedding.weight.device
        # token ids and the EOT positions (EOT is the largest id) are index bookkeeping: prepared on the host when the tokens come from
        # clip.tokenize (a CPU tensor), so no cast / reduce kernel runs on the device for them
        if text.is_cuda:
            tokens = text.to(torch.int32).contiguous()
            eot = tokens.argmax(dim=-1).to(torch.int32)
        else:
            t32 = text.to(torch.int32).contiguous()
            tokens, eot = t32.to(dev, non_blocking=True), t32.argmax(dim=-1).to(torch.int32).to(dev, non_blocking=True)
        n, L = tokens.shape
        D = self.token_embedding.weight.shape[1]
        dt = self.compute_dtype
        x = torch.empty((n, L, D), dtype=dt, device=dev)
        pos_res = getattr(self, "positional_embedding_res", self.positional_embedding)
        lib().call("egm_text_embed", dtype_code(dt), ptr(tokens), ptr(self.token_embedding.weight.detach()), ptr(self.positional_embedding.detach()),
                   ptr(pos_res.detach()), 20, ptr(x), n, L, D, stream())
        for blk in self.transformer.resblocks:
            x = blk.run(x, "causal")
        x = O.layernorm(x, self.ln_final)
        sel = torch.empty((n, D), dtype=dt, device=dev)
        lib().call("egm_gather_rows", dtype_code(dt), ptr(x), ptr(eot), ptr(sel), n, L, D, stream())
        return O.matmul_kn(sel, self.text_projection).float()

    @torch.no_grad()
    def encode_image(self, image, return_all=False, csa=True):
        require_gpu()
        feat, _ = self.visual.run(image.to(self.visual.conv1.weight.device), self.compute_dtype, csa_all_layers=False)
        return feat.float()


def build_model(state_dict: dict, load_from_clip: bool = False):
    """Same shape inference as the reference's build_model (clip/model.py:654-691), including its fp16 round trip of the
    GEMM weights (convert_weights, :631-652) so that results match a reference-loaded checkpoint."""
    sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    vw = sd["visual.conv1.weight"].shape[0]
    vlayers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    embed_dim = sd["text_projection"].shape[1]
    tw = sd["ln_final.weight"].shape[0]
    tlayers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")})
    model = CLIP(embed_dim, patch * grid, vlayers, vw, patch, sd["positional_embedding"].shape[0], sd["token_embedding.weight"].shape[0], tw, tw // 64,
                 tlayers, load_from_clip)
    model.load_state_dict(sd)
    with torch.no_grad():                                             # fp16 round trip of exactly the tensors convert_weights touches
        for m in model.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                m.weight.copy_(m.weight.half().float())
                if m.bias is not None:
                    m.bias.copy_(m.bias.half().float())
            if isinstance(m, nn.MultiheadAttention):
                m.in_proj_weight.copy_(m.in_proj_weight.half().float()); m.in_proj_bias.copy_(m.in_proj_bias.half().float())
        model.text_projection.copy_(model.text_projection.half().float())
        model.visual.proj.copy_(model.visual.proj.half().float())
    return model.eval()
