"""CLIPSeg (CLIPDensePredT) on the CLIP fork with the reference's constructor / forward / state_dict surface
(models/clipseg.py:136-332, 359-496), inference path on the HIP library.

    model = CLIPDensePredT(version='ViT-B/16', reduce_dim=64)       # loads weights/longclip-B.pt when present
    model.load_state_dict(torch.load('weights/rd64-uni.pth'), strict=False)
    mask_logits = model(images, prompts)[0]                          # [B, 1, H, W] fp32
"""
import math
import os

import torch
import torch.nn as nn

from ._lib import dtype_code, lib, ptr, require_gpu, stream
from .clip import ops as O
from .clip.model import CLIP, build_model
from .clip.tokenizer import tokenize

_VIT = {"ViT-B/16": dict(patch=16, token_shape=(14, 14)), "ViT-B/32": dict(patch=32, token_shape=(7, 7))}


def get_prompt_list(prompt):
    if prompt == "plain":
        return ["{}"]
    if prompt == "fixed":
        return ["a photo of a {}."]
    if prompt == "shuffle":
        return ["a photo of a {}.", "a photograph of a {}.", "an image of a {}.", "{}."]
    if prompt == "shuffle+":
        return ["a photo of a {}.", "a photograph of a {}.", "an image of a {}.", "{}.", "a cropped photo of a {}.", "a good photo of a {}.",
                "a photo of one {}.", "a bad photo of a {}.", "a photo of the {}."]
    raise ValueError("Invalid value for prompt")


class CLIPDensePredT(nn.Module):
    def __init__(self, version="ViT-B/32", extract_layers=(3, 6, 9), cond_layer=0, reduce_dim=128, n_heads=4, prompt="fixed", extra_blocks=0,
                 reduce_cond=None, fix_shift=False, learn_trans_conv_only=False, limit_to_clip_only=False, upsample=False,
                 add_calibration=False, rev_activations=False, trans_conv=None, n_tokens=None, complex_trans_conv=False,
                 clip_weights="weights/longclip-B.pt"):
        super().__init__()
        for flag, name in ((extra_blocks, "extra_blocks"), (reduce_cond, "reduce_cond"), (fix_shift, "fix_shift"), (upsample, "upsample"),
                           (n_tokens, "n_tokens"), (complex_trans_conv, "complex_trans_conv"), (trans_conv, "trans_conv")):
            if flag:
                raise NotImplementedError(f"egm_unet_amd: CLIPDensePredT({name}=...) is not used by the reference scripts and not implemented")
        cfg = _VIT[version]
        if os.path.isfile(clip_weights):                            # models/clipseg.py:147 (hard-coded relative path there)
            self.clip_model = build_model(torch.load(clip_weights, map_location="cpu"), load_from_clip=False)
        else:                                                       # the reference ships no weights: random-init backbone of the same shape
            self.clip_model = CLIP(512, 224, 12, 768, cfg["patch"], 248, 49408, 512, 8, 12, load_from_clip=False).eval()
        self.model = self.clip_model.visual
        self.n_tokens = None
        for p in self.clip_model.parameters():
            p.requires_grad_(False)
        self.reduce_cond = None
        self.film_mul = nn.Linear(512, reduce_dim)
        self.film_add = nn.Linear(512, reduce_dim)
        self.reduce = nn.Linear(768, reduce_dim)
        self.prompt_list = get_prompt_list(prompt)
        self.precomputed_prompts = dict()
        self.extract_layers, self.cond_layer = extract_layers, cond_layer
        self.limit_to_clip_only, self.process_cond, self.rev_activations = limit_to_clip_only, None, rev_activations
        self.upsample_proj, self.add_activation1, self.version = None, True, version
        self.token_shape = cfg["token_shape"]
        self.shift_vector = None
        ks = (cfg["patch"], cfg["patch"])
        self.trans_conv = nn.ConvTranspose2d(reduce_dim, 1, ks, stride=ks)
        depth = len(extract_layers)
        self.reduces = nn.ModuleList([nn.Linear(768, reduce_dim) for _ in range(depth)])
        self.blocks = nn.ModuleList([nn.TransformerEncoderLayer(d_model=reduce_dim, nhead=n_heads) for _ in range(depth)])
        self.extra_blocks = nn.ModuleList([])
        self.n_heads = n_heads
        self.compute_dtype = torch.float32
        self.decoder_dropout = None          # None: the encoder layers' own p (0.1, as in the reference's train mode); 0.0 disables it

    def set_compute_dtype(self, dtype):
        self.clip_model.set_compute_dtype(dtype)
        self.compute_dtype = dtype
        return self

    # ---- conditionals (models/clipseg.py:266-332)
    @torch.no_grad()
    def compute_conditional(self, conditional):
        dev = next(self.parameters()).device
        if type(conditional) in {list, tuple}:
            # (the token tensor stays on the host: encode_text prepares ids and EOT positions there -- no cast / argmax kernels on the device)
            return self.clip_model.encode_text(tokenize(list(conditional), context_length=248, truncate=True))
        if conditional in self.precomputed_prompts:
            return self.precomputed_prompts[conditional].float().to(dev)
        return self.clip_model.encode_text(tokenize([conditional], context_length=248, truncate=True))[0]

    def get_cond_vec(self, conditional, batch_size):
        if conditional is not None and type(conditional) == str:
            return self.compute_conditional(conditional).repeat(batch_size, 1)
        if conditional is not None and type(conditional) in {list, tuple} and type(conditional[0]) == str:
            assert len(conditional) == batch_size
            return self.compute_conditional(conditional)
        if conditional is not None and type(conditional) == torch.Tensor and conditional.ndim == 2:
            return conditional
        if conditional is not None and type(conditional) == torch.Tensor:
            return self.visual_forward(conditional)[0]
        raise ValueError("invalid conditional")

    @torch.no_grad()
    def visual_forward(self, x_inp, extract_layers=(), skip=False, mask=None):
        """-> (visual_q [B, 512] fp32, activations [L, B, 768] fp32 like the reference, affinities [] (not materialised)).
        mask = (layer | 'all', 'cls_token', seg [B, H, W]): the visual-prompt mask of CLIPDensePredTMasked (models/clipseg.py:222-231):
        seg is sampled (nearest, like nnf.interpolate's default) onto the token grid and multiplies the class token's attention row
        in the selected layers."""
        q, acts = self._visual_run(x_inp, extract_layers, mask)
        return q.float(), [a.float().permute(1, 0, 2) for a in acts], []

    @torch.no_grad()
    def _visual_run(self, x_inp, extract_layers=(), mask=None):
        """The encoder pass behind visual_forward in the compute dtype, batch-first: (q [B, 512], [activations [B, L, 768]]).  forward() takes
        these as they are; the reference's fp32 [L, B, 768] copies (4 x 48 MB per call at B = 32) are only made where they are returned."""
        require_gpu()
        dev = self.model.conv1.weight.device
        cls_mask = None
        if mask is not None:
            mask_layer, mask_type, seg = mask
            if mask_type != "cls_token":
                raise NotImplementedError("egm_unet_amd: visual_forward masks of type 'cls_token' only (what CLIPDensePredTMasked uses)")
            g = x_inp.shape[2] // self.model.patch_size
            seg = seg.to(dev).float()
            iy = (torch.arange(g, device=dev) * seg.shape[1] // g).long()          # nearest source index = floor(dst * in / out)
            ix = (torch.arange(g, device=dev) * seg.shape[2] // g).long()
            cls_mask = (mask_layer, seg[:, iy][:, :, ix].reshape(seg.shape[0], g * g).contiguous())
        return self.model.run(x_inp.to(dev), self.compute_dtype, extract_layers=tuple(extract_layers), cls_mask=cls_mask)

    def _encoder_layer(self, blk, a):
        """nn.TransformerEncoderLayer defaults in eval mode: post-norm, ReLU feed-forward, no dropout."""
        qkv = O.linear(a, blk.self_attn.in_proj_weight, blk.self_attn.in_proj_bias)
        att = O.attention(qkv, self.n_heads, "full")
        a = O.layernorm(O.linear(att, blk.self_attn.out_proj.weight, blk.self_attn.out_proj.bias, residual=a), blk.norm1)
        h = O.linear(a, blk.linear1.weight, blk.linear1.bias, act=1)
        return O.layernorm(O.linear(h, blk.linear2.weight, blk.linear2.bias, residual=a), blk.norm2)

    def forward(self, inp_image, conditional=None, return_features=False, mask=None):
        """models/clipseg.py:436-496.  eval(): inference path, no autograd.  train(): the decoder (reduces, FiLM, blocks, trans_conv)
        is differentiable through the HIP autograd operators of clip/train_ops.py; the CLIP backbone stays frozen (:155-156);
        nn.TransformerEncoderLayer's dropout is applied (self.decoder_dropout overrides its p; 0.0 switches it off)."""
        assert type(return_features) == bool
        if mask is not None:
            raise ValueError("mask not supported")                  # as the reference (models/clipseg.py:442-443)
        if self.training and torch.is_grad_enabled():
            return self._forward_train(inp_image, conditional, return_features)
        with torch.no_grad():
            return self._forward_eval(inp_image, conditional, return_features)

    def _encoder_layer_train(self, blk, a):
        """nn.TransformerEncoderLayer in train mode (post-norm, ReLU): dropout p on the attention weights, behind the attention output
        projection, behind the ReLU and behind the second feed-forward linear (models/clipseg.py:421-422, p = 0.1 by default)."""
        from .clip import train_ops as T
        p = self.decoder_dropout if self.decoder_dropout is not None else float(blk.dropout.p)
        qkv = T.linear(a, blk.self_attn.in_proj_weight, blk.self_attn.in_proj_bias)
        att = T.attention(qkv, self.n_heads, p)
        if p == 0.0:
            a = T.layernorm(T.linear(att, blk.self_attn.out_proj.weight, blk.self_attn.out_proj.bias, residual=a), blk.norm1)
            h = T.linear(a, blk.linear1.weight, blk.linear1.bias, act=1)
            return T.layernorm(T.linear(h, blk.linear2.weight, blk.linear2.bias, residual=a), blk.norm2)
        a = T.layernorm(T.dropout(T.linear(att, blk.self_attn.out_proj.weight, blk.self_attn.out_proj.bias), p, residual=a), blk.norm1)
        h = T.dropout(T.linear(a, blk.linear1.weight, blk.linear1.bias, act=1), p)
        return T.layernorm(T.dropout(T.linear(h, blk.linear2.weight, blk.linear2.bias), p, residual=a), blk.norm2)

    def _forward_train(self, inp_image, conditional, return_features):
        from .clip import train_ops as T
        dev = self.model.positional_embedding.device
        x_inp = inp_image.to(dev)
        bs = x_inp.shape[0]
        with torch.no_grad():
            cond = self.get_cond_vec(conditional, bs)
            q_raw, acts_all = self._visual_run(x_inp, extract_layers=[0] + list(self.extract_layers))
        dt, code = self.compute_dtype, dtype_code(self.compute_dtype)
        acts = acts_all[1:]
        acts = acts[::-1] if not self.rev_activations else acts
        condT = torch.empty(cond.shape, dtype=dt, device=dev)
        lib().call("egm_cast_f32", code, ptr(cond.float().contiguous()), ptr(condT), cond.numel(), stream())
        a = None
        for i, (act, blk, red) in enumerate(zip(acts, self.blocks, self.reduces)):
            a = T.linear(act.detach(), red.weight, red.bias, residual=a)
            if i == self.cond_layer:
                a = T.FilmFn.apply(a, T.linear(condT, self.film_mul.weight, self.film_mul.bias),
                                   T.linear(condT, self.film_add.weight, self.film_add.bias))
            a = self._encoder_layer_train(blk, a)
        out = T.TransConvFn.apply(a, self.trans_conv.weight, self.trans_conv.bias)
        if return_features:
            return out, q_raw.float(), cond, [t.float().permute(1, 0, 2) for t in acts_all]
        return out,

    def _forward_eval(self, inp_image, conditional=None, return_features=False):
        dev = self.model.positional_embedding.device
        x_inp = inp_image.to(dev)
        bs = x_inp.shape[0]
        cond = self.get_cond_vec(conditional, bs)
        q_raw, acts_all = self._visual_run(x_inp, extract_layers=[0] + list(self.extract_layers))
        dt, code, L_ = self.compute_dtype, dtype_code(self.compute_dtype), lib()
        acts = acts_all[1:]
        acts = acts[::-1] if not self.rev_activations else acts
        condT = torch.empty(cond.shape, dtype=dt, device=dev)
        L_.call("egm_cast_f32", code, ptr(cond.float().contiguous()), ptr(condT), cond.numel(), stream())
        a = None
        for i, (act, blk, red) in enumerate(zip(acts, self.blocks, self.reduces)):
            a = O.linear(act, red.weight, red.bias, residual=a)
            if i == self.cond_layer:
                mul, add = O.linear(condT, self.film_mul.weight, self.film_mul.bias), O.linear(condT, self.film_add.weight, self.film_add.bias)
                L_.call("egm_film", code, ptr(a), ptr(mul), ptr(add), bs, a.shape[1], a.shape[2], stream())
            a = self._encoder_layer(blk, a)
        Ltot, rd = a.shape[1], a.shape[2]
        g = int(math.isqrt(Ltot - 1))
        P = self.trans_conv.kernel_size[0]
        y = torch.empty((bs * Ltot, P * P), dtype=dt, device=dev)      # per-token 64 -> 16x16 patch (ConvTranspose2d as a GEMM)
        O.gemm(a.reshape(-1, rd), rd, O.cast_weight(self.trans_conv.weight.reshape(rd, P * P), dt), P * P, False, y, P * P, bs * Ltot, P * P, rd, dt)
        out = torch.empty((bs, 1, g * P, g * P), dtype=torch.float32, device=dev)
        L_.call("egm_pixel_shuffle", code, ptr(y), P * P, 1, Ltot, ptr(self.trans_conv.bias.detach().float()), ptr(out), bs, g, P, stream())
        if return_features:
            return out, q_raw.float(), cond, [t.float().permute(1, 0, 2) for t in acts_all]
        return out,


class CLIPDensePredTMasked(CLIPDensePredT):
    """CLIPSeg conditioned on a support image + its segmentation (models/clipseg.py:500-525): the conditional vector is the CLIP
    image feature of the support image, computed with the class token's attention restricted to the masked region in every layer."""

    def __init__(self, version="ViT-B/32", extract_layers=(3, 6, 9), cond_layer=0, reduce_dim=128, n_heads=4, prompt="fixed", extra_blocks=0,
                 reduce_cond=None, fix_shift=False, learn_trans_conv_only=False, refine=None, limit_to_clip_only=False, upsample=False,
                 add_calibration=False, n_tokens=None, **kw):
        super().__init__(version=version, extract_layers=extract_layers, cond_layer=cond_layer, reduce_dim=reduce_dim, n_heads=n_heads,
                         prompt=prompt, extra_blocks=extra_blocks, reduce_cond=reduce_cond, fix_shift=fix_shift,
                         learn_trans_conv_only=learn_trans_conv_only, limit_to_clip_only=limit_to_clip_only, upsample=upsample,
                         add_calibration=add_calibration, n_tokens=n_tokens, **kw)

    def visual_forward_masked(self, img_s, seg_s):
        return super().visual_forward(img_s, mask=("all", "cls_token", seg_s))

    def forward(self, img_q, cond_or_img_s, seg_s=None, return_features=False):
        if seg_s is None:
            cond = cond_or_img_s
        else:
            with torch.no_grad():
                cond, _, _ = self.visual_forward_masked(cond_or_img_s, seg_s)
        return super().forward(img_q, cond, return_features=return_features)
