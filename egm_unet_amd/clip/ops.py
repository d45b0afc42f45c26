"""Inference operators of the CLIP / CLIPSeg path over libegm_hip.so (no autograd: the CLIP backbone is frozen in the
reference, models/clipseg.py:155-156, and only forward passes are used by predict_CLIPseg.py / eval_CLIPseg.py)."""
import torch

from .._lib import dtype_code, lib, ptr, stream

_cast_cache = {}
_cast_generation = [0]


def bump_cast_generation():
    """Called by optimizers that update parameters through raw pointers (no torch version bump)."""
    _cast_generation[0] += 1


def cast_weight(w: torch.Tensor, dtype):
    """fp32 parameter -> contiguous matrix in the activation dtype (cached on storage + version)."""
    if dtype == torch.float32:
        return w.detach().contiguous()
    # a view of a parameter (weight.reshape(...), a slice of in_proj_weight) is a new tensor object on every call: the cache entry hangs on
    # the base tensor + offset + shape, so those hit too (they re-cast every step before: 17 cast launches per CLIPSeg forward)
    base = w._base if w._base is not None else w
    slot = (id(base), w.storage_offset(), tuple(w.shape), tuple(w.stride()))
    key = (w.data_ptr(), base._version, _cast_generation[0], dtype)
    hit = _cast_cache.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    src = w.detach().contiguous()
    out = torch.empty(src.shape, dtype=dtype, device=w.device)
    lib().call("egm_cast_f32", dtype_code(dtype), ptr(src), ptr(out), src.numel(), stream())
    _cast_cache[slot] = (key, out)
    return out


def gemm(A, lda, B, ldb, transB, C, ldc, M, N, K, dtype, bias=None, act=0, R=None, ldr=0, alpha=1.0, c_f32=False, nb1=1, nb2=1,
         sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), offA=0, offB=0, offC=0):
    """Thin wrapper over egm_gemm; A/B/C/R are tensors, off* element offsets into them."""
    es = 2 if dtype == torch.bfloat16 else 4
    ces = 4 if c_f32 else es
    import ctypes
    pa = ctypes.c_void_p(A.data_ptr() + offA * es)
    pb = ctypes.c_void_p(B.data_ptr() + offB * es)
    pc = ctypes.c_void_p(C.data_ptr() + offC * ces)
    lib().call("egm_gemm", dtype_code(dtype), pa, lda, pb, ldb, 1 if transB else 0, pc, ldc, 1 if c_f32 else 0, ptr(bias), act, ptr(R), ldr,
               float(alpha), M, N, K, nb1, nb2, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], sR[0], sR[1], stream())


def linear(x, weight, bias=None, act=0, residual=None):
    """x [..., K] (dtype T) @ weight[N, K]^T (+bias) -> [..., N]; act: 0 none, 1 ReLU, 2 QuickGELU; residual added after act."""
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    w = cast_weight(weight, x.dtype)
    b = bias.detach().float().contiguous() if bias is not None else None
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    r2 = residual.reshape(-1, N) if residual is not None else None
    gemm(x2, K, w, K, True, out, N, M, N, K, x.dtype, bias=b, act=act, R=r2, ldr=N)
    return out.reshape(*x.shape[:-1], N)


def matmul_kn(x, w_kn):
    """x [M, K] @ w[K, N] (weight stored [K][N], e.g. visual.proj / text_projection)."""
    M, K = x.shape
    N = w_kn.shape[1]
    w = cast_weight(w_kn, x.dtype)
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    gemm(x, K, w, N, False, out, N, M, N, K, x.dtype)
    return out


def layernorm(x, ln):
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    y = torch.empty_like(x2)
    lib().call("egm_layernorm", dtype_code(x.dtype), ptr(x2), D, ptr(ln.weight.detach().float()), ptr(ln.bias.detach().float()), float(ln.eps),
               ptr(y), D, x2.shape[0], D, stream())
    return y.reshape(x.shape)


def attention(qkv, n_heads, mode, cls_mask=None):
    """qkv [B, L, 3D] -> [B, L, D].  mode: 'csa' (softmax(qq^T s) + softmax(kk^T s)), 'causal', 'full'.
    cls_mask [nmask, L-1] fp32: multiplies the class token's attention row, head bh taking row bh % nmask (the reference's pairing,
    models/clipseg.py:111-117); runs on the unfused path (the probabilities are materialised)."""
    B, L, D3 = qkv.shape
    D = D3 // 3
    dh = D // n_heads
    Lp = (L + 7) // 8 * 8
    dt, dev = qkv.dtype, qkv.device
    if dt == torch.bfloat16 and dh == 64 and qkv.is_contiguous() and cls_mask is None:
        # fused path: scores and probabilities stay on chip (csrc/vit.hip attention_fused_kernel)
        out = torch.empty((B, L, D), dtype=dt, device=dev)
        lib().call("egm_attention_fused", dtype_code(dt), ptr(qkv), D3, B, L, n_heads, dh, {"full": 0, "causal": 1, "csa": 2}[mode], ptr(out), D,
                   stream())
        return out
    S = torch.empty((B * n_heads, L, Lp), dtype=torch.float32, device=dev)
    P = torch.empty((B * n_heads, L, Lp), dtype=dt, device=dev)
    scale = dh ** -0.5
    L_, code = lib(), dtype_code(dt)

    def scores(off_a, off_b):
        gemm(qkv, D3, qkv, D3, True, S, Lp, L, L, dh, dt, alpha=scale, c_f32=True, nb1=B, nb2=n_heads, sA=(L * D3, dh), sB=(L * D3, dh),
             sC=(n_heads * L * Lp, L * Lp), offA=off_a, offB=off_b)

    if mode == "csa":
        scores(0, 0)
        L_.call("egm_softmax_rows", code, ptr(S), Lp, ptr(P), Lp, B * n_heads * L, L, 0, 0, stream())
        scores(D, D)
        L_.call("egm_softmax_rows", code, ptr(S), Lp, ptr(P), Lp, B * n_heads * L, L, 0, 1, stream())
    else:
        scores(0, D)
        L_.call("egm_softmax_rows", code, ptr(S), Lp, ptr(P), Lp, B * n_heads * L, L, 1 if mode == "causal" else 0, 0, stream())
    if cls_mask is not None:
        m = cls_mask.float().contiguous()
        if m.shape[1] != L - 1:
            raise RuntimeError(f"attention: class-token mask has {m.shape[1]} entries, the sequence {L - 1} patch tokens")
        L_.call("egm_attn_mask_cls", code, ptr(P), Lp, L * Lp, ptr(m), m.shape[0], B * n_heads, L - 1, stream())
    out = torch.empty((B, L, D), dtype=dt, device=dev)
    gemm(P, Lp, qkv, D3, False, out, D, L, dh, L, dt, nb1=B, nb2=n_heads, sA=(n_heads * L * Lp, L * Lp), sB=(L * D3, dh), sC=(L * D, dh), offB=2 * D)
    return out
