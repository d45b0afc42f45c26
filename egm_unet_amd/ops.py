"""Autograd operators over the HIP library: every arithmetic step of the model goes through libegm_hip.so.

Activations are NHWC tensors ``[N, H, W, C]`` (fp32 or bf16, C a multiple of 8, possibly a channel-slice view of a
wider buffer).  torch is used for memory (torch.empty / views), streams and autograd bookkeeping only.
"""
import ctypes
import os
import struct
import weakref

import torch
from torch.autograd import Function, Variable

from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, dtype_code, lib, ptr, stream  # noqa: F401


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def _nhwc(t: torch.Tensor):
    """-> (tensor usable by the kernels, ld).  Accepts dense NHWC tensors and channel-slice views of them."""
    assert t.dim() == 4, t.shape
    N, H, W, C = t.shape
    s = t.stride()
    ok = (s[3] == 1 and s[2] % 8 == 0 and s[2] >= C and s[1] == W * s[2] and s[0] == H * W * s[2]
          and t.data_ptr() % 16 == 0 and C % 8 == 0)
    if not ok:
        if C % 8:
            raise RuntimeError(f"egm_unet_amd: channel count {C} is not a multiple of 8")
        t = t.contiguous()
        s = t.stride()
    return t, s[2]


def _npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


def _f32(n, device, zero=False):
    return (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=device)


_zero_pools = {}


def _zero_grad_vec(n, device):
    """An all-zero fp32 gradient of n elements for a parameter whose gradient is identically zero (a conv bias in front of a train-mode
    BatchNorm): a slice of a zero buffer per device, allocated OUTSIDE any capture, instead of a fill kernel per parameter and step
    (14 launches per EGM-UNet step).  Every caller gets storage of its own: when a buffer is used up the next one is allocated (the
    slices handed out keep the old one alive), never a slice that somebody else already holds -- a user-side in-place write to such a
    gradient (p.grad.add_(), foreach optimizers with weight decay + maximize, gradient noise) then touches that one bias only.
    A captured graph never refills its slices, so writes to them persist across replays: use them read-only under graph replay."""
    pool = _zero_pools.get(device)
    if pool is None or pool[0].numel() < pool[1] + n:
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros(n, dtype=torch.float32, device=device)      # the need arises inside a capture: plain fill this time
        pool = _zero_pools[device] = [torch.zeros(max(1 << 16, 4 * n), dtype=torch.float32, device=device), 0]
    off = pool[1]
    pool[1] = off + n
    return pool[0][off:off + n]


def _bn_conv_bias_grad(Cout, dy, training, device):
    """Gradient of a conv bias in front of a BatchNorm: identically zero when the BatchNorm normalises with batch statistics (the mean
    subtraction removes the bias), sum(dy) over the pixels when it is frozen (eval mode with grad enabled)."""
    return _zero_grad_vec(Cout, device) if training else _channel_sum(dy)[0, :Cout]


# ----------------------------------------------------------------------------------------------------------
# layout conversion at the module boundary
# ----------------------------------------------------------------------------------------------------------
class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, x_nchw, dtype):
        x = x_nchw.contiguous().float()
        N, C, H, W = x.shape
        out = torch.empty((N, H, W, pad8(C)), dtype=dtype, device=x.device)
        lib().call("egm_nchw_to_nhwc", dtype_code(dtype), ptr(x), ptr(out), pad8(C), N, C, H, W, stream())
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, g):
        g, ld = _nhwc(g)
        N, H, W, _ = g.shape
        out = torch.empty((N, ctx.C, H, W), dtype=torch.float32, device=g.device)
        lib().call("egm_nhwc_to_nchw", dtype_code(g.dtype), ptr(g), ld, ptr(out), N, ctx.C, H, W, stream())
        return out, None


class _ToNCHW(Function):
    @staticmethod
    def forward(ctx, x, C):
        x, ld = _nhwc(x)
        N, H, W, CP = x.shape
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
        lib().call("egm_nhwc_to_nchw", dtype_code(x.dtype), ptr(x), ld, ptr(out), N, C, H, W, stream())
        ctx.dtype, ctx.CP = x.dtype, CP
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        N, C, H, W = g.shape
        out = torch.empty((N, H, W, ctx.CP), dtype=ctx.dtype, device=g.device)
        lib().call("egm_nchw_to_nhwc", dtype_code(ctx.dtype), ptr(g), ptr(out), ctx.CP, N, C, H, W, stream())
        return out, None


def to_nhwc(x_nchw, dtype):
    return _ToNHWC.apply(x_nchw, dtype)


def to_nchw(x_nhwc, C):
    return _ToNCHW.apply(x_nhwc, C)


# ----------------------------------------------------------------------------------------------------------
# descriptor tables for the multi-tensor kernels
# ----------------------------------------------------------------------------------------------------------
# Pointer tables are uploaded through pinned staging buffers.  A captured hipGraph re-copies the SAME pinned bytes on every replay, so
# the tables a capture uses must never be rewritten by anybody else (an eager step between replays, another capture): everything
# issued under table_namespace(tag) -- GraphedTrainStep runs its warm-up and its capture under one tag -- gets buffers of its own.
_table_tag = [None]


class table_namespace:
    def __init__(self, tag):
        self.tag, self.prev = tag, None

    def __enter__(self):
        self.prev, _table_tag[0] = _table_tag[0], self.tag

    def __exit__(self, *a):
        _table_tag[0] = self.prev


_all_tables = weakref.WeakSet()


def drop_table_namespace(tag):
    """Forget the buffers every DeviceTable holds for `tag` (a captured graph that no longer exists)."""
    for t in list(_all_tables):
        t.slots.pop(tag, None)


class DeviceTable:
    """Packed C structs on the device.  Re-uploaded only when the bytes change; the upload is an async copy from a pinned
    staging buffer, so it is legal inside hipGraph capture (replays re-copy the same bytes).  One set of buffers per table namespace.

    Inside a capture a table may be needed with SEVERAL contents (the data-parallel step flushes the deferred weight gradients once
    per captured backward half): a replay re-reads the pinned bytes at replay time, so each distinct content gets a buffer pair of its
    own from a small pool reserved before the capture (pinned allocation is not legal while capturing)."""
    POOL = 4

    def __init__(self):
        self.slots = {}                      # tag -> [key, pinned, device, upload event, spare pairs, {blob: device} of this capture]
        _all_tables.add(self)

    def _slot(self):
        return self.slots.setdefault(_table_tag[0], [None, None, None, None, [], {}])

    def reserve(self, device, n=65536):
        """Allocate the staging buffers now (pinned allocation is not legal inside a hipGraph capture)."""
        sl = self._slot()
        if sl[1] is None or sl[1].numel() < n:
            cap = max(n, 65536)
            sl[1] = torch.empty(cap, dtype=torch.uint8).pin_memory()
            sl[2] = torch.empty(cap, dtype=torch.uint8, device=device)
            sl[0] = sl[3] = None
            sl[4] = [(torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=device))
                     for _ in range(self.POOL - 1)] if _table_tag[0] is not None else []
            sl[5] = {}

    def get(self, blob: bytes, device):
        sl = self._slot()
        if blob == sl[0]:
            return sl[2]
        n = len(blob)
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # inside a capture the primary pair is never rewritten (it may serve a content captured earlier): every further content
            # takes a spare pair, whose pinned bytes are written once, here
            hit = sl[5].get(blob)
            if hit is not None:
                return hit
            if not sl[4] or sl[4][0][0].numel() < n:
                raise RuntimeError("egm_unet_amd: table pool exhausted inside a hipGraph capture (DeviceTable.POOL)")
            pinned, dev_buf = sl[4].pop(0)
            pinned[:n] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            dev_buf[:n].copy_(pinned[:n], non_blocking=True)
            sl[5][blob] = dev_buf
            sl.append((pinned, dev_buf))                       # the pair lives as long as the namespace
            return dev_buf
        self.reserve(device, n)
        upload_pinned(sl, 1, 2, 3, torch.frombuffer(bytearray(blob), dtype=torch.uint8), n)
        sl[0] = blob
        return sl[2]


def upload_pinned(slot, i_pinned, i_device, i_event, host_values, n):
    """Rewrite the first n elements of a pinned staging buffer and copy them to its device twin (async).  The host write must not
    overtake the previous upload from the same buffer (the copy engine may not have read it yet when the host runs ahead of the
    GPU), so every upload records an event and the next rewrite waits for it.  Inside a hipGraph capture neither is possible nor
    needed: a capture owns its table namespace, whose buffers are written once."""
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    ev = slot[i_event]
    if ev is not None and not capturing:
        ev.synchronize()
    slot[i_pinned][:n] = host_values
    slot[i_device][:n].copy_(slot[i_pinned][:n], non_blocking=True)
    if not capturing and slot[i_device].is_cuda:
        if ev is None:
            ev = slot[i_event] = torch.cuda.Event()
        ev.record()


# ----------------------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------------------
_pack_cache = {}
_weight_generation = [0]


def bump_weight_generation():
    """Called by optimizers that update parameters through raw pointers (no torch version bump)."""
    _weight_generation[0] += 1


def _packed_weights(weight, groups, dtype):
    """fp32 OIHW parameter -> (wf, wd) operand packs; cached on (storage, version) so eval loops don't repack."""
    key = (weight.data_ptr(), weight._version, _weight_generation[0], dtype, groups, tuple(weight.shape))
    hit = _pack_cache.get(id(weight))
    if hit is not None and hit[0] == key and hit[3]() is weight:    # the weakref guards against id()/address reuse by a new tensor
        return hit[1], hit[2]
    Cout, Cin_g, KH, KW = weight.shape
    Cin = Cin_g * groups
    wf = torch.empty((KH * KW, pad8(Cout), pad8(Cin)), dtype=dtype, device=weight.device)
    wd = torch.empty((KH * KW, pad8(Cin), pad8(Cout)), dtype=dtype, device=weight.device)
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    lib().call("egm_conv_pack", dtype_code(dtype), ptr(w), ptr(wf), ptr(wd), Cout, Cin, KH, KW, groups, stream())
    _pack_cache[id(weight)] = (key, wf, wd, weakref.ref(weight))
    return wf, wd


def prepack_model(model, dtype):
    """Pack every nn.Conv2d weight of `model` (except holders flagged _egm_no_prepack) in ONE kernel launch."""
    st = getattr(model, "_egm_prepack", None)
    if st is None or st["dtype"] != dtype:
        convs = [m for m in model.modules() if isinstance(m, torch.nn.Conv2d) and not getattr(m, "_egm_no_prepack", False)
                 and m.weight.dim() == 4 and m.weight.is_cuda]
        bufs = []
        for m in convs:
            Cout, Cin_g, KH, KW = m.weight.shape
            Cin = Cin_g * m.groups
            bufs.append((torch.empty((KH * KW, pad8(Cout), pad8(Cin)), dtype=dtype, device=m.weight.device),
                         torch.empty((KH * KW, pad8(Cin), pad8(Cout)), dtype=dtype, device=m.weight.device)))
        st = model._egm_prepack = {"dtype": dtype, "convs": convs, "bufs": bufs, "table": DeviceTable(), "stamp": None}
    convs, bufs = st["convs"], st["bufs"]
    if not convs:
        return
    stamp = (_weight_generation[0], tuple(m.weight._version for m in convs), tuple(m.weight.data_ptr() for m in convs))
    if stamp == st["stamp"]:
        return
    L = lib()
    chunk = L.cdll.egm_conv_pack_chunk()
    blob, chunks = bytearray(), 0
    for m, (wf, wd) in zip(convs, bufs):
        Cout, Cin_g, KH, KW = m.weight.shape
        Cin = Cin_g * m.groups
        blob += struct.pack("<QQQiiiiiiii", m.weight.data_ptr(), wf.data_ptr(), wd.data_ptr(), Cout, Cin, pad8(Cout), pad8(Cin), KH, KW,
                            m.groups, chunks)                  # last field = index of the entry's first workgroup
        chunks += (KH * KW * pad8(Cout) * pad8(Cin) + chunk - 1) // chunk
    dev = convs[0].weight.device
    table = st["table"].get(bytes(blob), dev)
    L.call("egm_conv_pack_multi", dtype_code(dtype), ptr(table), len(convs), chunks, stream())
    for m, (wf, wd) in zip(convs, bufs):
        w = m.weight
        key = (w.data_ptr(), w._version, _weight_generation[0], dtype, m.groups, tuple(w.shape))
        _pack_cache[id(w)] = (key, wf, wd, weakref.ref(w))
    st["stamp"] = stamp


# deferred weight-gradient reductions: the slab kernels run inside backward, ONE multi-conv reduce runs when backward ends.
# Deferral hands autograd a gradient tensor that is filled later, which is only sound when that tensor is the weight's ONLY
# contribution in this backward (autograd would otherwise sum unfilled buffers): _conv_uses counts the conv forwards a weight has
# taken part in since its gradients were last produced, and a weight used more than once takes the immediate path.
_pending_wgrad = []
_wgrad_run = [None]                       # graph-task id of the backward run whose end-of-run flush is queued
_conv_uses = {}                           # id(weight) -> [weakref, weight generation, forwards awaiting their backward, peak of that]


def _note_conv_use(weight):
    ent = _conv_uses.get(id(weight))
    if ent is None or ent[0]() is not weight or ent[1] != _weight_generation[0]:
        ent = _conv_uses[id(weight)] = [weakref.ref(weight), _weight_generation[0], 0, 0]
        if len(_conv_uses) > 4096:                                    # drop entries of dead tensors
            for k in [k for k, v in _conv_uses.items() if v[0]() is None]:
                del _conv_uses[k]
    ent[2] += 1
    ent[3] = max(ent[3], ent[2])


def _sole_conv_use(weight):
    """True when `weight` fed exactly one conv forward that still awaits its backward (the one running now); consumes that use."""
    ent = _conv_uses.get(id(weight))
    if ent is None or ent[0]() is not weight:
        return False
    sole = ent[3] == 1
    ent[2] -= 1
    if ent[2] <= 0:
        del _conv_uses[id(weight)]
    return sole
_wgrad_table = DeviceTable()
_wgrad_table_partial = DeviceTable()      # mid-backward flushes (eager gradient exchange): never disturbs the bytes a captured graph re-uploads


def _flush_wgrads(ready_only=False):
    """Finish the deferred weight gradients with ONE multi-conv reduction.  Runs as an autograd-engine callback when backward ends;
    ready_only=True (a gradient bucket is about to be gathered mid-backward, parallel.GradAllReducer) finishes the convs whose
    gradient tensor autograd has already adopted and leaves the others pending."""
    if not ready_only:
        _wgrad_run[0] = None
    _flush_bgrads(ready_only)
    _launch_pending_slabs()
    if not _pending_wgrad:
        return
    todo, later = [], []
    for ent in _pending_wgrad:
        weight, gref, gptr = ent[1], ent[9], ent[10]
        # The gradient tensor returned from backward() was handed over to autograd with no reference kept here, so that it is
        # adopted as weight.grad without a copy (backward()) or returned to the caller as is (torch.autograd.grad()).  It is still
        # reachable either through the weak reference or as the storage weight.grad now shares.
        g = gref()
        if g is None and weight.grad is not None and weight.grad.data_ptr() == gptr:
            g = weight.grad
        if g is not None:
            todo.append(ent)
        elif ready_only:
            later.append(ent)
        elif weight.grad is not None:
            # the unfilled buffer is gone but the parameter HAS a gradient: autograd copied or transformed the buffer instead of
            # adopting it (a tensor hook returning a new tensor, a non-stealable gradient), so weight.grad holds whatever the
            # uninitialised buffer contained.  _conv_wgrad refuses to defer in the cases it can see; anything else must not train on.
            raise RuntimeError("deferred conv weight gradient lost its destination: weight.grad exists but is not the buffer "
                               "backward() returned (tensor hook / gradient copy); set EGM_DEFER_WGRAD=0 for this model")
        # else: nobody holds the gradient and the parameter has none (torch.autograd.grad() result dropped, or the weight was not
        # among backward(inputs=...)): it was discarded, nothing to finish
    _pending_wgrad[:] = later
    if not todo:
        return
    blob, chunks = bytearray(), 0
    per = lib().cdll.egm_wgrad_reduce_chunk()
    for ws, weight, nslab, taps, CoutP, CinP, Cout, Cin, groups, gref, gptr in todo:
        blob += struct.pack("<QQiiiiiiiiii", ws.data_ptr(), gptr, nslab, taps, CoutP, CinP, Cout, Cin, groups, 0, chunks, 0)
        chunks += (taps * CoutP * CinP + per - 1) // per
    dev = todo[0][1].device
    _wgrad_table.reserve(dev); _wgrad_table_partial.reserve(dev)      # both exist before any capture can need them
    table = (_wgrad_table_partial if ready_only else _wgrad_table).get(bytes(blob), dev)
    lib().call("egm_wgrad_reduce_multi", ptr(table), len(todo), chunks, stream())


def flush_ready_wgrads():
    _flush_wgrads(ready_only=True)


# deferred bias gradients: db = sum over pixels of dy for every nn.Conv2d bias that no BatchNorm follows, as ONE pair of launches when
# backward ends (egm_bias_grad_multi) instead of a pair per layer inside it.  Same hand-over protocol as the deferred weight gradients:
# backward() returns an unfilled [Cout] tensor that autograd adopts as bias.grad (no reference kept here, so it is not copied), the
# queue entry keeps dy alive and finds the tensor again through a weak reference or bias.grad's address.
_DEFER_BGRAD = os.environ.get("EGM_DEFER_BGRAD", "1") != "0"
_pending_bgrad = []
_bgrad_table = DeviceTable()
_bgrad_table_partial = DeviceTable()
_BSUM_ENTRY = struct.Struct("<3Qq6i")       # egm_bsum_entry


def defer_bgrads(enabled=None):
    """Get / set whether conv bias gradients are left to the one multi-tensor pass at the end of backward."""
    global _DEFER_BGRAD
    if enabled is not None:
        _DEFER_BGRAD = bool(enabled)
    return _DEFER_BGRAD


def _bias_grad(gy, Cout, bias=None, defer=False):
    """db[c] = sum over pixels of gy[..., c] (gy NHWC, channels padded) -> fp32 [Cout].  defer: the conv's weight gradient is being
    deferred in this backward (so this is the only use of the layer in it) -- the bias gradient then joins the end-of-backward pass."""
    if (defer and _DEFER_BGRAD and bias is not None and bias.is_leaf and bias.grad is None and not bias._backward_hooks
            and not torch.is_grad_enabled()):
        gy, ldg = _nhwc(gy)
        gb = torch.empty(Cout, dtype=torch.float32, device=gy.device)
        _wgrad_run_begin()
        _pending_bgrad.append((gy, ldg, _npix(gy), gy.shape[3], Cout, bias, weakref.ref(gb), gb.data_ptr()))
        return gb
    return _channel_sum(gy)[0, :Cout]


def _flush_bgrads(ready_only=False):
    if not _pending_bgrad:
        return
    todo, later = [], []
    for ent in _pending_bgrad:
        bias, gref, gptr = ent[5], ent[6], ent[7]
        g = gref()
        if g is None and bias.grad is not None and bias.grad.data_ptr() == gptr:
            g = bias.grad
        if g is not None:
            todo.append(ent)
        elif ready_only:
            later.append(ent)
        elif bias.grad is not None:
            raise RuntimeError("deferred conv bias gradient lost its destination: bias.grad exists but is not the buffer backward() "
                               "returned (tensor hook / gradient copy); set EGM_DEFER_BGRAD=0 for this model")
        # else: the gradient was discarded (torch.autograd.grad() result dropped, bias not among backward(inputs=...))
    _pending_bgrad[:] = later
    if not todo:
        return
    L = lib()
    by_dtype = {}
    for ent in todo:
        by_dtype.setdefault(ent[0].dtype, []).append(ent)
    for dtype, ents in by_dtype.items():
        dev = ents[0][0].device
        nblks = [L.query("egm_channel_partials_blocks", npix, C) for _gy, _ld, npix, C, *_ in ents]
        ws = torch.empty(sum(nb * 2 * ent[3] for nb, ent in zip(nblks, ents)), dtype=torch.float32, device=dev)
        blob1, blob2, b1, b2, off = bytearray(), bytearray(), 0, 0, 0
        for nb, (gy, ldg, npix, C, Cout, _bias, _gref, gptr) in zip(nblks, ents):
            part = ws.data_ptr() + 4 * off
            blob1 += _BSUM_ENTRY.pack(gy.data_ptr(), part, gptr, npix, ldg, C, Cout, nb, b1, 0)
            blob2 += _BSUM_ENTRY.pack(gy.data_ptr(), part, gptr, npix, ldg, C, Cout, nb, b2, 0)
            b1 += nb; b2 += C // 8; off += nb * 2 * C
        _bgrad_table.reserve(dev); _bgrad_table_partial.reserve(dev)
        table = (_bgrad_table_partial if ready_only else _bgrad_table).get(bytes(blob1 + blob2), dev)
        L.call("egm_bias_grad_multi", dtype_code(dtype), ptr(table), len(ents), b1, b2, stream())


def _channel_sum(t):
    """sum over pixels of an NHWC tensor -> fp32 [2, C] (row 0 = sum, row 1 = sum of squares)."""
    t, ld = _nhwc(t)
    C, npix = t.shape[3], _npix(t)
    nb = lib().query("egm_channel_partials_blocks", npix, C)
    part = _f32(nb * 2 * C, t.device)
    out = _f32((2, C), t.device)
    lib().call("egm_channel_sums", dtype_code(t.dtype), ptr(t), ld, npix, C, ptr(part), stream())
    lib().call("egm_reduce_tiles", ptr(part), nb, C, ptr(out), stream())
    return out


_DEFER_WGRAD = os.environ.get("EGM_DEFER_WGRAD", "1") != "0"

class Lazy:
    """A logical activation z = act(scale*y + shift) that has NOT been written to memory: `y` is the raw conv output (the autograd
    stand-in: its gradient is, by convention, dL/dz), `coef` the fp32 [4, C] rows scale | shift | mean | rstd of the BatchNorm.
    Handed to a consumer whose first pass applies the BatchNorm itself and writes the tensor on the way (the MCALayer's statistics pass,
    the classifier inside the apply pass, the fused skip-connection pool); everything else calls materialize().
    (Rounds 2-3 also let convolutions consume it through operand prologues; measured slower three times -- DESIGN.md 6.2, 6.4, 6.5 --
    and removed in round 4.)"""
    __slots__ = ("y", "coef", "act")

    def __init__(self, y, coef, act):
        self.y, self.coef, self.act = y, coef, act

    shape = property(lambda self: self.y.shape)
    dtype = property(lambda self: self.y.dtype)
    device = property(lambda self: self.y.device)

    def materialize(self, out=None):
        return _Materialize.apply(self.y, self.coef, self.act, None if out is None else [out])


def materialize(x, out=None):
    """Lazy -> tensor (optionally into the destination view `out`); tensors pass through (copied only if `out` is another place)."""
    if isinstance(x, Lazy):
        return x.materialize(out)
    if out is not None and not (x.data_ptr() == out.data_ptr() and x.stride() == out.stride()):
        return _axpby(x, 1.0, None, 0.0, out)
    return x


class _Materialize(Function):
    """z = act(scale*y + shift) as a tensor.  The gradient of the stand-in y IS dL/dz (see Lazy), so backward is the identity:
    the activation derivative and the BatchNorm backward both happen in the producing _ConvBN node."""

    @staticmethod
    def forward(ctx, y, coef, act, out_slot):
        y, ldy = _nhwc(y)
        N, H, W, CP = y.shape
        z, ldz = _slot_or_new(out_slot, (N, H, W, CP), y.dtype, y.device)
        lib().call("egm_bn_act_fwd", dtype_code(y.dtype), ptr(y), ldy, ptr(coef[0]), ptr(coef[1]), act, ptr(z), ldz, _npix(y), CP, stream())
        return z

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


def _unlazy(x):
    """-> (tensor, coef or None, act): a consumer's memory operand and the BatchNorm + activation it has to apply to it"""
    if isinstance(x, Lazy):
        return x.y, x.coef, x.act
    return x, None, ACT_NONE


_GROUP_CONVS = os.environ.get("EGM_GROUP_CONVS", "1") != "0"


def group_convs(enabled=None):
    """Get / set whether conv_group() merges launches (tests compare both ways)."""
    global _GROUP_CONVS
    if enabled is not None:
        _GROUP_CONVS = bool(enabled)
    return _GROUP_CONVS


class conv_group:
    """`with conv_group():` -- the egm_conv_fwd* launches issued inside are recorded by the library and launched together on exit, those
    of one kernel instantiation as ONE launch (csrc/group.h).  Only for convolutions that are independent of each other and whose
    outputs are first used after the block.  EGM_GROUP_CONVS=0 turns it into a no-op."""

    depth = 0                                 # open conv_group blocks, whether or not merging is switched on (bench.py's bookkeeping)
    serial = 0                                # number of the outermost block that is open / was opened last

    def __enter__(self):
        self.on = _GROUP_CONVS
        conv_group.depth += 1
        if conv_group.depth == 1:
            conv_group.serial += 1
        if self.on:
            lib().call("egm_group_begin")
        return self

    def __exit__(self, et, ev, tb):
        conv_group.depth -= 1
        if self.on:
            if et is None:
                lib().call("egm_group_end", stream())
            else:
                lib().cdll.egm_group_abort()
        return False


def _conv_forward(x, ldx, weight, bias, dil, groups, want_stats):
    """The conv launch shared by _Conv2d and _ConvBN: y = conv(x, weight) (+bias), optional BN partial statistics."""
    N, H, W, CinP = x.shape
    Cout, Cin_g, KH, KW = weight.shape
    Cin = Cin_g * groups
    if pad8(Cin) != CinP:
        raise RuntimeError(f"conv2d: input has {CinP} channels, weight expects {Cin} (padded {pad8(Cin)})")
    CoutP = pad8(Cout)
    L, dt = lib(), dtype_code(x.dtype)
    wf, wd = _packed_weights(weight, groups, x.dtype)
    y = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=x.device)
    stats = None
    if want_stats:
        ntiles = L.query("egm_conv_stats_tiles", dt, N, H, W, CinP, CoutP, KH, KW, dil)
        stats = _f32((ntiles, 2, CoutP), x.device)
    b = bias.detach() if bias is not None else None
    L.call("egm_conv_fwd", dt, ptr(x), ldx, ptr(wf), ptr(b), Cout if b is not None else 0, ptr(y), CoutP, ptr(stats), N, H, W, CinP, CoutP,
           KH, KW, dil, stream())
    return y, stats, wd


def _wgrad_deferrable(weight):
    """True when the slab reduction of this weight's gradient may be left to the ONE multi-conv launch at the end of backward: a leaf
    parameter whose only gradient contribution this is (not when a tensor hook may replace the returned buffer, nor under create_graph:
    both hand autograd something it copies).  Consumes the conv-use note of the weight: call once per backward of a conv."""
    return (_DEFER_WGRAD and _sole_conv_use(weight) and weight.is_leaf and weight.grad is None
            and not weight._backward_hooks and not torch.is_grad_enabled())


def _wgrad_run_begin():
    run = torch._C._current_graph_task_id()
    if run != _wgrad_run[0]:              # per engine run (an aborted backward never ran its callback: its entries are dead)
        _pending_wgrad.clear()
        _pending_slab_launch.clear()
        _pending_bgrad.clear()
        Variable._execution_engine.queue_callback(_flush_wgrads)
        _wgrad_run[0] = run


def _queue_wgrad(ws, weight, nslab, taps, CoutP, CinP, Cout, Cin, groups, gw):
    _wgrad_run_begin()
    _pending_wgrad.append((ws, weight, nslab, taps, CoutP, CinP, Cout, Cin, groups, weakref.ref(gw), gw.data_ptr()))


# ---- 1x1 convolutions: data gradient + weight-gradient slabs in ONE pass over dy and x (csrc/pw_bn.hip, egm_conv1x1_bwd) -----------
_FUSE_C1 = os.environ.get("EGM_CONV1X1_BWD", "1") != "0"
_C1_DESC = struct.Struct("<5Qq6i")
_C1_MAXC = int(os.environ.get("EGM_CONV1X1_BWD_MAXC", "64"))


def fuse_c1(enabled=None):
    """Get / set whether the backward of a 1x1 conv runs as one fused launch (dx + weight-gradient slabs) where that applies."""
    global _FUSE_C1
    if enabled is not None:
        _FUSE_C1 = bool(enabled)
    return _FUSE_C1


def _c1_shape_ok(x, gy, weight, dil, groups):
    # measured (profiles/r04_*): <= 64 channels on both sides 32-43 us against 60 us for the pair at 8 x 256^2 x 64; at 128 channels the
    # fused kernel needs one workgroup per CU and two column blocks and LOSES (73 us against 43 us), so those keep the pair
    return (_FUSE_C1 and weight.shape[2] == 1 and weight.shape[3] == 1 and groups == 1 and x.shape[3] <= _C1_MAXC and gy.shape[3] <= _C1_MAXC
            and bool(lib().cdll.egm_conv1x1_bwd_supported(dtype_code(x.dtype), x.shape[3], gy.shape[3])))


def _conv1x1_bwd(items):
    """items: up to 4 tuples (x, ldx, dy, lddy, weight, wd, need_gx, defer) of 1x1 convs -> [(gx, gw)].  ONE launch writes every dx and
    every slab set; the slabs of a deferrable weight join the deferred multi-conv reduction like _conv_wgrad's, the others are reduced
    at once.  (The kernel choice must not depend on `defer`: a weight is not deferrable when its conv's backward runs a second time --
    the split data-parallel step re-runs the bottleneck -- and both runs have to produce the same dx bit for bit.)"""
    L, st = lib(), stream()
    descs, res, now = [], [], []
    dt = dtype_code(items[0][0].dtype)
    for x, ldx, dy, lddy, weight, wd, need_gx, defer in items:
        N, H, W, CinP = x.shape
        CoutP, npix = dy.shape[3], _npix(x)
        Cout, Cin = weight.shape[0], weight.shape[1]
        nslab = L.query("egm_conv1x1_bwd_slabs", npix)
        ws = torch.empty(nslab * CoutP * CinP + 4, dtype=torch.float32, device=x.device)
        gx = torch.empty((N, H, W, CinP), dtype=x.dtype, device=x.device) if need_gx else None
        gw = torch.empty_like(weight)
        descs.append(_C1_DESC.pack(x.data_ptr(), dy.data_ptr(), wd.data_ptr(), 0 if gx is None else gx.data_ptr(), ws.data_ptr(), npix, ldx, lddy,
                                   CinP, CinP, CoutP, 0))
        if defer:
            _queue_wgrad(ws, weight, nslab, 1, CoutP, CinP, Cout, Cin, 1, gw)
        else:
            now.append((ws, gw, nslab, CoutP, CinP, Cout, Cin))
        res.append((gx, gw))
    L.call("egm_conv1x1_bwd", dt, b"".join(descs), len(descs), st)
    for ws, gw, nslab, CoutP, CinP, Cout, Cin in now:
        L.call("egm_wgrad_reduce", ptr(ws), ptr(gw), nslab, 1, CoutP, CinP, Cout, Cin, 1, 0, st)
    return res


def _conv_grads(x, ldx, dy, weight, wd, dil, groups, Cin, Cout, need_gx, need_gw, x_split=0):
    """(data gradient, weight gradient) of a conv from a materialised dy [N, H, W, CoutP] (contiguous): one fused launch for a 1x1 conv
    whose weight gradient is deferrable, the weight-gradient slab kernel followed by the data-gradient conv otherwise."""
    N, H, W, CinP = x.shape
    CoutP = dy.shape[3]
    if need_gw and not x_split and _c1_shape_ok(x, dy, weight, dil, groups):
        return _conv1x1_bwd([(x, ldx, dy, CoutP, weight, wd, need_gx, _wgrad_deferrable(weight))])[0]
    gx = gw = None
    if need_gw:
        gw = _conv_wgrad(x, ldx, dy, CoutP, weight, dil, groups, Cin, Cout)
    if need_gx:
        gx = _conv_dgrad(dy, wd, N, H, W, CoutP, CinP, weight.shape[2], weight.shape[3], dil, x_split)
    return gx, gw


_MERGE_WGRAD = os.environ.get("EGM_MERGE_WGRAD", "1") != "0"
_pending_slab_launch = []                 # deferred slab-kernel launches (argument tuples holding their tensors)


def merge_wgrads(enabled=None):
    """Get / set whether the slab kernels of deferred weight gradients are launched together when backward ends (merged launches)."""
    global _MERGE_WGRAD
    if enabled is not None:
        _MERGE_WGRAD = bool(enabled)
    return _MERGE_WGRAD


_WGRAD_DESC = struct.Struct("<3Q14i")       # egm_conv_wgrad_desc
_wgrad_names = {}


def _wgrad_kernel_of(dt, N, H, W, CinP, CoutP, KH, KW, dil):
    key = (dt, N, H, W, CinP, CoutP, KH, KW, dil)
    name = _wgrad_names.get(key)
    if name is None:
        buf = ctypes.create_string_buffer(96)
        lib().cdll.egm_conv_wgrad_kernel_name(dt, N, H, W, CinP, CoutP, KH, KW, dil, ctypes.cast(buf, ctypes.c_void_p), 96)
        name = _wgrad_names[key] = buf.value
    return name


def _launch_pending_slabs():
    """The queued slab kernels, in queue order, those of one kernel instantiation four to a launch (egm_conv_wgrad_multi)."""
    if not _pending_slab_launch:
        return
    L, st = lib(), stream()
    todo = list(_pending_slab_launch)
    _pending_slab_launch.clear()
    buckets = {}
    for ent in todo:
        dt, x, ldx, gy, ldg, ws, N, H, W, CinP, CoutP, Cin, Cout, KH, KW, dil, groups = ent
        buckets.setdefault((dt, _wgrad_kernel_of(dt, N, H, W, CinP, CoutP, KH, KW, dil)), []).append(ent)
    per = 4                                                            # EGM_WGRAD_MULTI_MAX
    for (dt, _name), ents in buckets.items():
        for i in range(0, len(ents), per):
            blob = b"".join(_WGRAD_DESC.pack(x.data_ptr(), gy.data_ptr(), ws.data_ptr(), ldx, ldg, N, H, W, CinP, CoutP, Cin, Cout, KH, KW, dil, groups, 0)
                            for _dt, x, ldx, gy, ldg, ws, N, H, W, CinP, CoutP, Cin, Cout, KH, KW, dil, groups in ents[i:i + per])
            L.call("egm_conv_wgrad_multi", dt, blob, len(ents[i:i + per]), st)


def _conv_wgrad(x, ldx, gy, ldg, weight, dil, groups, Cin, Cout, defer=None):
    """Weight gradient of one conv; deferred slab reduction when that is safe."""
    N, H, W, CinP = x.shape
    CoutP = gy.shape[3]
    KH, KW = weight.shape[2], weight.shape[3]
    L, dt, st = lib(), dtype_code(x.dtype), stream()
    gw = torch.empty_like(weight)
    nbytes = L.query("egm_conv_wgrad_workspace", N, H, W, CinP, CoutP, KH, KW)
    ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=x.device)
    if defer is None:                     # (a caller that asked already -- _wgrad_deferrable consumes the use note -- passes the answer)
        defer = _wgrad_deferrable(weight)
    if defer and _MERGE_WGRAD:
        _wgrad_run_begin()
        # nothing reads the slabs before the end-of-backward reduction: the launch itself waits for it too and shares launches with the
        # other deferred ones (_flush_wgrads); x and gy stay alive in the queue entry
        _pending_slab_launch.append((dt, x, ldx, gy, ldg, ws, N, H, W, CinP, CoutP, Cin, Cout, KH, KW, dil, groups))
    else:
        L.call("egm_conv_wgrad", dt, ptr(x), ldx, ptr(gy), ldg, None if defer else ptr(gw), ptr(ws), N, H, W, CinP, CoutP, Cin, Cout, KH, KW,
               dil, groups, 0, st)
    if defer:
        nslab = L.query("egm_conv_wgrad_slabs", dt, N, H, W, CinP, CoutP, KH, KW, dil)
        _queue_wgrad(ws, weight, nslab, KH * KW, CoutP, CinP, Cout, Cin, groups, gw)
    return gw


# ---- deferred dz: a BatchNorm backward that computes its incoming gradient from its producer's inputs (csrc/bn_dz_fused.hip) -------
# The producer's backward (the 1x1 classifier's data gradient; the MCALayer's last step) does not run its kernel: it returns a STAND-IN
# tensor of the right shape and registers what the BatchNorm backward needs under the stand-in's address; _ConvBN.backward picks it up.
# Only used where the model code knows the tensor between the two nodes has exactly one consumer (so autograd hands the stand-in through
# unchanged); an entry nobody consumed by the end of the backward pass is an error, not a silent garbage gradient.
_FUSE_DZ = os.environ.get("EGM_FUSE_DZ", "1") != "0"
_DEFERRED_DZ = {}


def fuse_dz(enabled=None):
    """Get / set whether BatchNorm backward computes dz on the fly from the classifier / MCALayer behind it (tests compare both ways)."""
    global _FUSE_DZ
    if enabled is not None:
        _FUSE_DZ = bool(enabled)
    return _FUSE_DZ


def _check_deferred_dz():
    _dz_run[0] = None
    if _DEFERRED_DZ:
        kinds = [v[1][0] for v in _DEFERRED_DZ.values()]
        _DEFERRED_DZ.clear()
        raise RuntimeError(f"egm_unet_amd: deferred BatchNorm gradients {kinds} were never consumed (the tensor between the producer and "
                           "its BatchNorm had another consumer); call ops.fuse_dz(False)")


_dz_run = [None]                          # autograd graph-task id of the backward run whose end-of-run check is queued


def _defer_dz(standin, payload):
    # The check is per ENGINE RUN, not per "registry was empty": the engine drops its callbacks when a backward raises (OOM, launch
    # error, KeyboardInterrupt), so entries of an aborted run would otherwise stay for the life of the process, pin their payload
    # tensors, and keep the check from ever being queued again.  A new graph-task id means a new run: stale entries are dropped
    # (their stand-ins belong to a backward that no longer exists) and this run gets its own check.
    run = torch._C._current_graph_task_id()
    if run != _dz_run[0]:
        _DEFERRED_DZ.clear()
        Variable._execution_engine.queue_callback(_check_deferred_dz)
        _dz_run[0] = run
    _DEFERRED_DZ[standin.data_ptr()] = (standin, payload)          # the stand-in stays alive: its address cannot be recycled meanwhile


def _dz_fusable(C):
    return _FUSE_DZ and C % 8 == 0 and C <= 1024 and 256 % (C // 8) == 0


class _Conv2d(Function):
    """nn.Conv2d (stride 1, same padding) on NHWC activations; weight stays the fp32 OIHW nn.Parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil, groups, want_stats, bias_grad_zero=False, defer_dgrad=False):
        x, ldx = _nhwc(x)
        Cout, Cin_g = weight.shape[0], weight.shape[1]
        y, stats, wd = _conv_forward(x, ldx, weight, bias, dil, groups, want_stats)
        if ctx.needs_input_grad[1]:
            _note_conv_use(weight)
        ctx.save_for_backward(x, weight, wd)
        ctx.meta = (dil, groups, bias is not None, Cin_g * groups, Cout)
        ctx.bias_grad_zero = bias_grad_zero
        ctx.bias = bias                                 # (a leaf parameter: looked at in backward for the deferred bias gradient)
        # the caller vouches that x is the materialised output of a conv -> BatchNorm(+act) node and has no other consumer
        ctx.defer_dgrad = bool(defer_dgrad and weight.shape[2] == 1 and weight.shape[3] == 1 and groups == 1
                               and Cout <= 8 and weight.is_contiguous() and _dz_fusable(x.shape[3]))
        ctx.set_materialize_grads(False)                # no zero-filled "gradient" for the non-differentiable stats output
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return (None,) * 8
        x, weight, wd = ctx.saved_tensors
        dil, groups, has_bias, Cin, Cout = ctx.meta
        gy, ldg = _nhwc(gy)
        x, ldx = _nhwc(x)
        N, H, W, CinP = x.shape
        CoutP = gy.shape[3]
        KH, KW = weight.shape[2], weight.shape[3]
        L, dt, st = lib(), dtype_code(x.dtype), stream()
        gx = gw = gb = None
        need_gx, need_gw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        defer = _wgrad_deferrable(weight) if need_gw else False     # (consumes the weight's use note: asked once per backward)
        if has_bias and ctx.needs_input_grad[2]:
            gb = _zero_grad_vec(Cout, x.device) if ctx.bias_grad_zero else _bias_grad(gy, Cout, ctx.bias, defer)
        if need_gw and not (need_gx and ctx.defer_dgrad and _FUSE_DZ) and _c1_shape_ok(x, gy, weight, dil, groups):
            # 1x1: data gradient + weight-gradient slabs from ONE pass over gy and x
            gx, gw = _conv1x1_bwd([(x, ldx, gy, ldg, weight, wd, need_gx, defer)])[0]
            return gx, gw, gb, None, None, None, None, None
        if need_gx:
            gx = torch.empty((N, H, W, CinP), dtype=x.dtype, device=x.device)
            if ctx.defer_dgrad and _FUSE_DZ:
                # never written: the BatchNorm backward in front computes dz = gy * W per vector itself (egm_bn_cls_bwd_*)
                _defer_dz(gx, ("cls", gy, ldg, weight.detach(), Cout, Cin))
            else:
                L.call("egm_conv_fwd", dt, ptr(gy), ldg, ptr(wd), None, 0, ptr(gx), CinP, None, N, H, W, CoutP, CinP, KH, KW, dil, st)
        if need_gw and gw is None:
            gw = _conv_wgrad(x, ldx, gy, ldg, weight, dil, groups, Cin, Cout, defer=defer)
        return gx, gw, gb, None, None, None, None, None


def conv2d(x, weight, bias=None, dil=1, groups=1, want_stats=False, bias_grad_zero=False, defer_dgrad=False):
    """x: NHWC tensor (a Lazy is materialised first).
    defer_dgrad: x is the materialised result of ops.conv_bn_act and feeds nothing else -- a 1x1 conv to <= 8 channels (the classifier)
    then leaves its data gradient to that BatchNorm's backward (see _defer_dz)."""
    return _Conv2d.apply(materialize(x), weight, bias, dil, groups, want_stats, bias_grad_zero, defer_dgrad)


def _conv_dgrad(dy, wd, N, H, W, CoutP, CinP, KH, KW, dil, x_split):
    """Data gradient of a conv from dy [N,H,W,CoutP] (contiguous).  x_split > 0: written as two dense tensors (the gradients of the
    two concatenated inputs) when the kernel supports it; the returned tensor is then a never-written stand-in registered for the
    concat's backward."""
    L, dt, st = lib(), dtype_code(dy.dtype), stream()
    gx = torch.empty((N, H, W, CinP), dtype=dy.dtype, device=dy.device)
    if x_split and _FUSE_DZ and L.cdll.egm_conv_split_ok(dt, N, H, W, CoutP, CinP, KH, KW, dil, x_split):
        ga = torch.empty((N, H, W, x_split), dtype=dy.dtype, device=dy.device)
        gb = torch.empty((N, H, W, CinP - x_split), dtype=dy.dtype, device=dy.device)
        L.call("egm_conv_fwd_split", dt, ptr(dy), CoutP, ptr(wd), ptr(ga), x_split, ptr(gb), CinP - x_split, x_split, N, H, W, CoutP, CinP,
               KH, KW, dil, st)
        _defer_dz(gx, ("split", ga, gb))
    else:
        L.call("egm_conv_fwd", dt, ptr(dy), CoutP, ptr(wd), None, 0, ptr(gx), CinP, None, N, H, W, CoutP, CinP, KH, KW, dil, st)
    return gx


_FUSE_CLS = os.environ.get("EGM_FUSE_CLS", "1") != "0"


def fuse_cls(enabled=None):
    """Get / set whether the 1x1 classifier runs inside the BatchNorm apply pass of the layer in front (tests compare both ways)."""
    global _FUSE_CLS
    if enabled is not None:
        _FUSE_CLS = bool(enabled)
    return _FUSE_CLS


class _BnActCls(Function):
    """Lazy (conv -> BatchNorm -> act stand-in) -> 1x1 classifier -> fp32 NCHW logits as ONE node (OutConv behind up4's DoubleConv,
    src/EGM-UNet.py:952-956 + :1536-1540): the BatchNorm apply pass writes z (kept for the classifier's weight gradient) and computes
    the logits from it (egm_bn_act_cls_fwd).  backward: classifier weight / bias gradients from (z, dlogits); the data gradient is
    left to the BatchNorm backward in front (_defer_dz "cls") or, with that switched off, computed by the conv kernel."""

    @staticmethod
    def forward(ctx, y, coef, act, weight, bias):
        y, ldy = _nhwc(y)
        N, H, W, C = y.shape
        nc, Cin = weight.shape[0], weight.shape[1]
        L, dt, st, dev = lib(), dtype_code(y.dtype), stream(), y.device
        z = torch.empty((N, H, W, C), dtype=y.dtype, device=dev)
        logits = torch.empty((N, nc, H, W), dtype=torch.float32, device=dev)
        L.call("egm_bn_act_cls_fwd", dt, ptr(y), ldy, ptr(coef[0]), ptr(coef[1]), act, ptr(z), C, ptr(weight.detach()), nc, Cin,
               ptr(bias.detach()) if bias is not None else None, ptr(logits), N, H, W, C, st)
        if ctx.needs_input_grad[3]:
            _note_conv_use(weight)
        ctx.save_for_backward(z, weight)
        ctx.meta = (nc, Cin, bias is not None)
        ctx.bias = bias
        return logits

    @staticmethod
    def backward(ctx, g):
        z, weight = ctx.saved_tensors
        nc, Cin, has_bias = ctx.meta
        N, H, W, C = z.shape
        L, dt, st, dev = lib(), dtype_code(z.dtype), stream(), z.device
        g = g.contiguous().float()
        dl = torch.empty((N, H, W, 8), dtype=z.dtype, device=dev)            # NHWC, channels nc..7 zero-filled
        L.call("egm_nchw_to_nhwc", dt, ptr(g), ptr(dl), 8, N, nc, H, W, st)
        gy = gw = gb = None
        defer = _wgrad_deferrable(weight) if ctx.needs_input_grad[3] else False
        if ctx.needs_input_grad[3]:
            gw = _conv_wgrad(z, C, dl, 8, weight, 1, 1, Cin, nc, defer=defer)
        if has_bias and ctx.needs_input_grad[4]:
            gb = _bias_grad(dl, nc, ctx.bias, defer)
        if ctx.needs_input_grad[0]:
            gy = torch.empty((N, H, W, C), dtype=z.dtype, device=dev)
            if _FUSE_DZ and _dz_fusable(C):
                _defer_dz(gy, ("cls", dl, 8, weight.detach(), nc, Cin))     # never written: see _defer_dz
            else:
                _, wd = _packed_weights(weight, 1, z.dtype)
                L.call("egm_conv_fwd", dt, ptr(dl), 8, ptr(wd), None, 0, ptr(gy), C, None, N, H, W, 8, C, 1, 1, 1, st)
        return gy, None, None, gw, gb


def bn_act_cls_ok(x, weight):
    """True when ops.bn_act_cls applies: x an ops.Lazy, a 1x1 classifier to <= 8 classes, channel count a power of two <= 512."""
    C = x.shape[3] if isinstance(x, Lazy) else 0
    return (_FUSE_CLS and isinstance(x, Lazy) and weight.dim() == 4 and weight.shape[2] == 1 and weight.shape[3] == 1 and weight.shape[0] <= 8
            and pad8(weight.shape[1]) == C and C % 8 == 0 and C // 8 <= 64 and (C // 8) & (C // 8 - 1) == 0 and weight.is_contiguous())


def bn_act_cls(x, weight, bias):
    """Lazy x -> fp32 NCHW logits of the 1x1 classifier (weight, bias); x's only consumer."""
    return _BnActCls.apply(x.y, x.coef, x.act, weight, bias)


class _ConvBN(Function):
    """conv -> BatchNorm (-> act, applied by whoever consumes the result) as ONE autograd node.

    forward : y = conv(x) with the BN statistics from the conv epilogue; egm_bn_finalize -> coef.  Returns the RAW conv output y as the
              stand-in of the logical z = act(scale*y + shift), and coef (see Lazy).
    backward: receives dL/dz (or a stand-in whose producer left dz to this node, see _defer_dz).  Partial sums (dz, y) ->
              egm_bn_bwd_coefs -> dy in one apply pass -> the conv's weight and data gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, act, training, dil, groups, x_split=0):
        """x_split = Cs > 0: x is a channel concatenation [Cs | rest] with no other consumer (ops.upcat): backward writes the gradient
        as two dense tensors where the kernel can (egm_conv_fwd_split) and hands them to the concat's backward (see _defer_dz)."""
        x, ldx = _nhwc(x)
        ctx.x_split = int(x_split) if (x_split and 0 < x_split < x.shape[3] and x_split % 8 == 0) else 0
        Cout, Cin_g = weight.shape[0], weight.shape[1]
        y, stats, wd = _conv_forward(x, ldx, weight, bias, dil, groups, training)
        CoutP, npix, dev = y.shape[3], _npix(y), y.device
        L, st = lib(), stream()
        coef = _f32((4, CoutP), dev)                    # scale, shift, save_mean, save_rstd
        if training:
            L.call("egm_bn_finalize", ptr(stats), stats.shape[0], npix, ptr(gamma.detach()), ptr(beta.detach()), eps, momentum,
                   ptr(running_mean), ptr(running_var), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), CoutP, Cout, st)
        else:
            L.call("egm_bn_eval_coeffs", ptr(gamma.detach()), ptr(beta.detach()), ptr(running_mean), ptr(running_var), eps,
                   ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), CoutP, Cout, st)
        if ctx.needs_input_grad[1]:
            _note_conv_use(weight)
        ctx.save_for_backward(x, weight, wd, y, coef)
        ctx.meta = (dil, groups, bias is not None, Cin_g * groups, Cout, act, training)
        ctx.mark_non_differentiable(coef)
        ctx.set_materialize_grads(False)
        return y, coef

    @staticmethod
    def backward(ctx, gz, _):
        if gz is None:
            return (None,) * 14
        x, weight, wd, y, coef = ctx.saved_tensors
        dil, groups, has_bias, Cin, Cout, act, training = ctx.meta
        pend = _DEFERRED_DZ.pop(gz.data_ptr(), None) if _DEFERRED_DZ else None      # gz is a stand-in: dz comes from its producer's inputs
        gz, ldg = _nhwc(gz)
        x, ldx = _nhwc(x)
        y, ldy = _nhwc(y)
        N, H, W, CinP = x.shape
        CoutP, npix, dev = y.shape[3], _npix(y), y.device
        L, dt, st = lib(), dtype_code(x.dtype), stream()
        scale, shift, mean, rstd = coef[0], coef[1], coef[2], coef[3]
        nb = L.query("egm_channel_partials_blocks", npix, CoutP)
        part = _f32(nb * 2 * CoutP, dev)
        sums, cf4 = _f32((2, CoutP), dev), _f32((4, CoutP), dev)
        train = 1 if training else 0
        need_gx, need_gw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dy = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=dev)
        if pend is not None:
            kind = pend[1][0]
            if kind == "cls":
                _, dl, lddl, wcls, nc, ldw = pend[1]
                head = ("egm_bn_cls", (dt, ptr(dl), lddl, ptr(wcls), nc, ldw), (npix, CoutP))
            else:
                _, dxo, ldd, gates, mcoef, ns, geo = pend[1]
                head = ("egm_bn_mca", (dt, ptr(dxo), ldd, ptr(gates), ptr(mcoef), ns), geo + (CoutP,))
            L.call(head[0] + "_bwd_reduce", *head[1], ptr(y), ldy, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act, ptr(part), *head[2], st)
            L.call("egm_bn_bwd_coefs", ptr(part), nb, npix, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), train, ptr(sums), ptr(cf4), CoutP, st)
            L.call(head[0] + "_bwd_apply", *head[1], ptr(y), ldy, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act, train, ptr(sums),
                   ptr(dy), CoutP, *head[2], st)
        else:
            L.call("egm_bn_act_bwd_reduce", dt, ptr(gz), ldg, ptr(y), ldy, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act, ptr(part),
                   npix, CoutP, st)
            L.call("egm_bn_bwd_coefs", ptr(part), nb, npix, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), train, ptr(sums),
                   ptr(cf4), CoutP, st)
            L.call("egm_bn_act_bwd_apply", dt, ptr(gz), ldg, ptr(y), ldy, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act,
                   train, ptr(sums), ptr(dy), CoutP, npix, CoutP, st)
        gx, gw = _conv_grads(x, ldx, dy, weight, wd, dil, groups, Cin, Cout, need_gx, need_gw, ctx.x_split)
        gb = _bn_conv_bias_grad(Cout, dy, training, dev) if (has_bias and ctx.needs_input_grad[2]) else None
        ggamma = sums[1, :Cout] if ctx.needs_input_grad[3] else None
        gbeta = sums[0, :Cout] if ctx.needs_input_grad[4] else None
        return gx, gw, gb, ggamma, gbeta, None, None, None, None, None, None, None, None, None


# environment switch for A/B runs: the max pool at a skip connection fused into its producer / consumer (csrc/pool_fused.hip)
_FUSE_POOL = os.environ.get("EGM_FUSE_POOL", "1") != "0"


_FUSE_MCA_BN = os.environ.get("EGM_FUSE_MCA_BN", "1") != "0"
_FUSE_MCA_BWD = os.environ.get("EGM_FUSE_MCA_BWD", "1") != "0"


def fuse_mca_bwd(enabled=None):
    """Get / set whether the MCALayer backward computes du and dxo in one tiled launch (bf16; egm_mca_bwd_dudxo)."""
    global _FUSE_MCA_BWD
    if enabled is not None:
        _FUSE_MCA_BWD = bool(enabled)
    return _FUSE_MCA_BWD


def fuse_mca_bn(enabled=None):
    """Get / set whether the BatchNorm+ReLU in front of an MCALayer is applied by the layer's statistics pass (egm_mca_reduce_bn)."""
    global _FUSE_MCA_BN
    if enabled is not None:
        _FUSE_MCA_BN = bool(enabled)
    return _FUSE_MCA_BN


def fuse_pool(enabled=None):
    """Get / set whether the skip-connection max pool is fused into the kernels on either side of it (tests compare both ways)."""
    global _FUSE_POOL
    if enabled is not None:
        _FUSE_POOL = bool(enabled)
    return _FUSE_POOL


class _ConvBNPool(Function):
    """conv -> BatchNorm -> act -> (z for the skip connection, maxpool2(z)) as ONE autograd node (the encoder's top level:
    src/EGM-UNet.py:44-55 followed by :908).  forward: the BatchNorm apply writes z and the pooled tensor in one pass
    (egm_bn_act_fwd_pool).  backward: receives the skip gradient and the pooled gradient; the BatchNorm backward passes compute
    dz = gskip + scatter(gpool) on the fly (egm_bn_pool_bwd_*), so the pool's scatter pass and the tensor it wrote do not exist."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, act, training, dil, groups, out_slot):
        x, ldx = _nhwc(x)
        Cout, Cin_g = weight.shape[0], weight.shape[1]
        y, stats, wd = _conv_forward(x, ldx, weight, bias, dil, groups, training)
        N, H, W, CoutP = y.shape
        npix, dev = _npix(y), y.device
        L, dt, st = lib(), dtype_code(y.dtype), stream()
        coef = _f32((4, CoutP), dev)                    # scale, shift, save_mean, save_rstd
        if training:
            L.call("egm_bn_finalize", ptr(stats), stats.shape[0], npix, ptr(gamma.detach()), ptr(beta.detach()), eps, momentum,
                   ptr(running_mean), ptr(running_var), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), CoutP, Cout, st)
        else:
            L.call("egm_bn_eval_coeffs", ptr(gamma.detach()), ptr(beta.detach()), ptr(running_mean), ptr(running_var), eps,
                   ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), CoutP, Cout, st)
        z, ldz = _slot_or_new(out_slot, (N, H, W, CoutP), y.dtype, dev)
        pooled = torch.empty((N, H // 2, W // 2, CoutP), dtype=y.dtype, device=dev)
        L.call("egm_bn_act_fwd_pool", dt, ptr(y), CoutP, ptr(coef[0]), ptr(coef[1]), act, ptr(z), ldz, ptr(pooled), CoutP, N, H, W, CoutP, st)
        if ctx.needs_input_grad[1]:
            _note_conv_use(weight)
        ctx.save_for_backward(x, weight, wd, y, coef, z)
        ctx.meta = (dil, groups, bias is not None, Cin_g * groups, Cout, act, training)
        ctx.set_materialize_grads(False)
        return z, pooled

    @staticmethod
    def backward(ctx, gz, gpool):
        if gz is None and gpool is None:
            return (None,) * 14
        x, weight, wd, y, coef, z = ctx.saved_tensors
        dil, groups, has_bias, Cin, Cout, act, training = ctx.meta
        x, ldx = _nhwc(x)
        N, H, W, CinP = x.shape
        CoutP, npix, dev = y.shape[3], _npix(y), y.device
        KH, KW = weight.shape[2], weight.shape[3]
        L, dt, st = lib(), dtype_code(x.dtype), stream()
        scale, shift, mean, rstd = coef[0], coef[1], coef[2], coef[3]
        sums, cf4 = _f32((2, CoutP), dev), _f32((4, CoutP), dev)
        dy = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=dev)
        train = 1 if training else 0
        if gz is not None and gpool is not None:
            gz, ldg = _nhwc(gz)
            gpool, ldp = _nhwc(gpool)
            nb = L.query("egm_bn_pool_bwd_blocks", N, H, W, CoutP)
            part = _f32(nb * 2 * CoutP, dev)
            L.call("egm_bn_pool_bwd_reduce", dt, ptr(gz), ldg, ptr(gpool), ldp, ptr(y), CoutP, ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                   act, ptr(part), N, H, W, CoutP, st)
            L.call("egm_bn_bwd_coefs", ptr(part), nb, npix, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), train, ptr(sums), ptr(cf4), CoutP, st)
            L.call("egm_bn_pool_bwd_apply", dt, ptr(gz), ldg, ptr(gpool), ldp, ptr(y), CoutP, ptr(scale), ptr(shift), ptr(mean), ptr(rstd),
                   act, train, ptr(sums), ptr(dy), CoutP, N, H, W, CoutP, st)
        else:
            # only one of the two consumers produced a gradient: the separate kernels
            if gz is None:
                zk, ldzk = _nhwc(z)
                gpool, ldp = _nhwc(gpool)
                gz = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=dev)
                L.call("egm_maxpool2_bwd", dt, ptr(zk), ldzk, ptr(gpool), ldp, ptr(gz), CoutP, N, H, W, CoutP, st)
            gz, ldg = _nhwc(gz)
            nb = L.query("egm_channel_partials_blocks", npix, CoutP)
            part = _f32(nb * 2 * CoutP, dev)
            L.call("egm_bn_act_bwd_reduce", dt, ptr(gz), ldg, ptr(y), CoutP, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act, ptr(part),
                   npix, CoutP, st)
            L.call("egm_bn_bwd_coefs", ptr(part), nb, npix, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), train, ptr(sums), ptr(cf4), CoutP, st)
            L.call("egm_bn_act_bwd_apply", dt, ptr(gz), ldg, ptr(y), CoutP, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act, train,
                   ptr(sums), ptr(dy), CoutP, npix, CoutP, st)
        gx = gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = _conv_wgrad(x, ldx, dy, CoutP, weight, dil, groups, Cin, Cout)
        if ctx.needs_input_grad[0]:
            gx = torch.empty((N, H, W, CinP), dtype=x.dtype, device=dev)
            L.call("egm_conv_fwd", dt, ptr(dy), CoutP, ptr(wd), None, 0, ptr(gx), CinP, None, N, H, W, CoutP, CinP, KH, KW, dil, st)
        if has_bias and ctx.needs_input_grad[2]:
            gb = _bn_conv_bias_grad(Cout, dy, training, dev)
        ggamma = sums[1, :Cout] if ctx.needs_input_grad[3] else None
        gbeta = sums[0, :Cout] if ctx.needs_input_grad[4] else None
        return gx, gw, gb, ggamma, gbeta, None, None, None, None, None, None, None, None, None


def conv_bn_lazy(x, conv, bn, act, dil=1, groups=1):
    """conv -> BatchNorm -> activation as an ops.Lazy: for a consumer whose first pass materialises the tensor itself (ops.mca_layer:
    BatchNorm apply + the three-axis statistics in one pass)."""
    if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_egm_counter_managed", False):
        bn.num_batches_tracked.add_(1)
    training = bn.training or bn.running_mean is None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    y, coef = _ConvBN.apply(materialize(x), conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, act,
                            training, dil, groups)
    return Lazy(y, coef, act)


def conv_bn_act_pool(x, conv, bn, act, dil=1, groups=1, out=None):
    """conv -> BatchNorm -> activation -> (result, maxpool2(result)): the skip-connection tensor (into the destination view `out` when
    given) and its pooled copy from one BatchNorm apply pass, one autograd node (see _ConvBNPool).  H and W must be even; callers fall
    back to conv_bn_act + fork_maxpool2 otherwise (pool_fusable)."""
    if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_egm_counter_managed", False):
        bn.num_batches_tracked.add_(1)                  # bookkeeping counter (int64), as nn.BatchNorm2d does
    training = bn.training or bn.running_mean is None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    x = materialize(x)
    return _ConvBNPool.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, act,
                             training, dil, groups, None if out is None else [out])


def pool_fusable(x):
    """True when the fused skip-connection pool applies to the NHWC tensor / Lazy x (switch on, even H and W, <= 1024 channels)."""
    return _FUSE_POOL and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and x.shape[1] >= 2 and x.shape[2] >= 2 and x.shape[3] <= 1024


# ----------------------------------------------------------------------------------------------------------
# BatchNorm (+ activation)
# ----------------------------------------------------------------------------------------------------------
class _BnAct(Function):
    @staticmethod
    def forward(ctx, y, stats, gamma, beta, running_mean, running_var, eps, momentum, act, training, out_slot=None):
        y, ldy = _nhwc(y)
        N, H, W, CP = y.shape
        C, npix, dev = gamma.shape[0], _npix(y), y.device
        L, dt, st = lib(), dtype_code(y.dtype), stream()
        coef = _f32((4, CP), dev)                       # scale, shift, save_mean, save_rstd
        scale, shift, mean, rstd = coef[0], coef[1], coef[2], coef[3]
        if training:
            if stats is None:
                nb = L.query("egm_channel_partials_blocks", npix, CP)
                stats = _f32((nb, 2, CP), dev)
                L.call("egm_channel_sums", dt, ptr(y), ldy, npix, CP, ptr(stats), st)
            L.call("egm_bn_finalize", ptr(stats), stats.shape[0], npix, ptr(gamma.detach()), ptr(beta.detach()), eps, momentum,
                   ptr(running_mean), ptr(running_var), ptr(scale), ptr(shift), ptr(mean), ptr(rstd), CP, C, st)
        else:
            L.call("egm_bn_eval_coeffs", ptr(gamma.detach()), ptr(beta.detach()), ptr(running_mean), ptr(running_var), eps,
                   ptr(scale), ptr(shift), ptr(mean), ptr(rstd), CP, C, st)
        z, ldz = _slot_or_new(out_slot, (N, H, W, CP), y.dtype, dev)
        L.call("egm_bn_act_fwd", dt, ptr(y), ldy, ptr(scale), ptr(shift), act, ptr(z), ldz, npix, CP, st)
        ctx.save_for_backward(y, coef)
        ctx.meta = (act, training, C)
        return z

    @staticmethod
    def backward(ctx, gz):
        y, coef = ctx.saved_tensors
        act, training, C = ctx.meta
        gz, ldg = _nhwc(gz)
        y, ldy = _nhwc(y)
        N, H, W, CP = y.shape
        npix, dev = _npix(y), y.device
        L, dt, st = lib(), dtype_code(y.dtype), stream()
        scale, shift, mean, rstd = coef[0], coef[1], coef[2], coef[3]
        nb = L.query("egm_channel_partials_blocks", npix, CP)
        part = _f32(nb * 2 * CP, dev)
        sums = _f32((2, CP), dev)
        L.call("egm_bn_act_bwd_reduce", dt, ptr(gz), ldg, ptr(y), ldy, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act,
               ptr(part), npix, CP, st)
        L.call("egm_reduce_tiles", ptr(part), nb, CP, ptr(sums), st)
        gy = None
        if ctx.needs_input_grad[0]:
            gy = torch.empty((N, H, W, CP), dtype=y.dtype, device=dev)
            L.call("egm_bn_act_bwd_apply", dt, ptr(gz), ldg, ptr(y), ldy, ptr(scale), ptr(shift), ptr(mean), ptr(rstd), act,
                   1 if training else 0, ptr(sums), ptr(gy), CP, npix, CP, st)
        ggamma = sums[1, :C] if ctx.needs_input_grad[2] else None
        gbeta = sums[0, :C] if ctx.needs_input_grad[3] else None
        return gy, None, ggamma, gbeta, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------------------
# several independent conv -> BatchNorm(+act) layers as one autograd node with multi-tensor BatchNorm passes
# ----------------------------------------------------------------------------------------------------------
BN_MULTI_MAX = 4
_BN_FINALIZE, _BN_FWD, _BN_BWD_REDUCE, _BN_BWD_COEFS, _BN_BWD_APPLY = range(5)    # enum egm_bn_multi_pass
_FUSE_BN_MULTI = os.environ.get("EGM_BN_MULTI", "1") != "0"


def _bn_desc(y=None, z=None, dz=None, dy=None, coef=None, stats=None, gamma=None, beta=None, rm=None, rv=None, partials=None, sums=None,
             cf4=None, npix=0, ldy=0, ldz=0, lddz=0, lddy=0, ntiles=0, nblocks=0, C=0, C_real=0, act=0, train=0, eps=0.0, momentum=0.0):
    """one packed egm_bn_desc (include/egm_hip.h)"""
    dp = lambda t: 0 if t is None else t.data_ptr()
    return struct.pack("<13Qq10i2f", dp(y), dp(z), dp(dz), dp(dy), dp(coef), dp(stats), dp(gamma), dp(beta), dp(rm), dp(rv), dp(partials),
                       dp(sums), dp(cf4), npix, ldy, ldz, lddz, lddy, ntiles, nblocks, C, C_real, act, train, eps, momentum)


class _MultiConvBN(Function):
    """K <= 4 INDEPENDENT conv -> BatchNorm -> activation layers (the parallel branches of EdgeEnhancedGRFB at equal depth) as one
    autograd node: the K convolutions are launched back to back, then ONE finalize launch and ONE apply launch serve all K BatchNorms;
    backward: ONE partial-sum launch, ONE coefficient launch, ONE apply launch, then the K data and weight gradients.  Every tensor
    goes through exactly the arithmetic of conv_bn_act (same kernels' bodies, same block decomposition), so results are bit-identical
    to K separate nodes; what disappears is 2 x (K - 1) forward and 3 x (K - 1) backward launches of 5 us each on tensors whose
    passes take about as long as a launch."""

    @staticmethod
    def forward(ctx, meta, *flat):
        K = len(meta)
        L, st = lib(), stream()
        saved, descs_fin, descs_fwd, outs, keep = [], b"", b"", [], []
        ctx.meta = []
        convs = []
        with conv_group():                                    # the K convolutions: one launch per kernel instantiation
            for k, mk in enumerate(meta):
                x, weight, bias = flat[5 * k:5 * k + 3]
                x, ldx = _nhwc(x)
                convs.append((x, ldx) + _conv_forward(x, ldx, weight, bias, mk[6], mk[7], mk[5]))
        for k, mk in enumerate(meta):
            x, weight, bias, gamma, beta = flat[5 * k:5 * k + 5]
            rm, rv, eps, momentum, act, training, dil, groups, out_slot = mk
            Cout, Cin_g = weight.shape[0], weight.shape[1]
            x, ldx, y, stats, wd = convs[k]
            CoutP, npix, dev = y.shape[3], _npix(y), y.device
            coef = _f32((4, CoutP), dev)
            if training:
                descs_fin += _bn_desc(coef=coef, stats=stats, gamma=gamma.detach(), beta=beta.detach(), rm=rm, rv=rv, npix=npix,
                                      ntiles=stats.shape[0], C=CoutP, C_real=Cout, eps=eps, momentum=momentum)
                keep.append(stats)
            else:
                L.call("egm_bn_eval_coeffs", ptr(gamma.detach()), ptr(beta.detach()), ptr(rm), ptr(rv), eps, ptr(coef[0]), ptr(coef[1]),
                       ptr(coef[2]), ptr(coef[3]), CoutP, Cout, st)
            z, ldz = _slot_or_new(out_slot, tuple(y.shape), y.dtype, dev)
            descs_fwd += _bn_desc(y=y, z=z, coef=coef, npix=npix, ldy=CoutP, ldz=ldz, C=CoutP, act=act)
            if ctx.needs_input_grad[1 + 5 * k + 1]:
                _note_conv_use(weight)
            saved += [x, weight, wd, y, coef]
            outs.append(z)
            ctx.meta.append((dil, groups, bias is not None, Cin_g * groups, Cout, act, training))
        dt = dtype_code(outs[0].dtype)
        if descs_fin:
            L.call("egm_bn_multi", dt, _BN_FINALIZE, descs_fin, len(descs_fin) // 160, st)
        L.call("egm_bn_multi", dt, _BN_FWD, descs_fwd, K, st)
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gz):
        K = len(ctx.meta)
        L, st = lib(), stream()
        sv = ctx.saved_tensors
        per, d_red, d_coef, d_app = [], b"", b"", b""
        for k in range(K):
            x, weight, wd, y, coef = sv[5 * k:5 * k + 5]
            dil, groups, has_bias, Cin, Cout, act, training = ctx.meta[k]
            g, ldg = _nhwc(gz[k])
            x, ldx = _nhwc(x)
            N, H, W, CinP = x.shape
            CoutP, npix, dev = y.shape[3], _npix(y), y.device
            nb = L.query("egm_channel_partials_blocks", npix, CoutP)
            part, sums, cf4 = _f32(nb * 2 * CoutP, dev), _f32((2, CoutP), dev), _f32((4, CoutP), dev)
            dy = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=dev)
            common = dict(y=y, dz=g, coef=coef, partials=part, sums=sums, cf4=cf4, npix=npix, ldy=CoutP, lddz=ldg, nblocks=nb, C=CoutP,
                          act=act, train=1 if training else 0)
            d_red += _bn_desc(**common)
            d_coef += _bn_desc(**common)
            d_app += _bn_desc(dy=dy, lddy=CoutP, **common)
            # every temporary stays referenced until the launches that use it are enqueued (freed earlier, the caching allocator
            # would hand its memory to the next layer's temporaries: the kernels are not in the stream yet)
            per.append((x, ldx, weight, wd, g, dy, sums, N, H, W, CinP, CoutP, part, cf4))
        dt = dtype_code(per[0][0].dtype)
        L.call("egm_bn_multi", dt, _BN_BWD_REDUCE, d_red, K, st)
        L.call("egm_bn_multi", dt, _BN_BWD_COEFS, d_coef, K, st)
        L.call("egm_bn_multi", dt, _BN_BWD_APPLY, d_app, K, st)
        grads = [None]
        gxs, gws = [None] * K, [None] * K
        # the 1x1 members (deferrable weight gradients): data gradient + weight-gradient slabs of all of them in ONE fused launch
        fused = [k for k in range(K) if ctx.needs_input_grad[1 + 5 * k + 1] and _c1_shape_ok(per[k][0], per[k][5], per[k][2], ctx.meta[k][0], ctx.meta[k][1])]
        if fused:
            res = _conv1x1_bwd([(per[k][0], per[k][1], per[k][5], per[k][11], per[k][2], per[k][3], ctx.needs_input_grad[1 + 5 * k],
                                 _wgrad_deferrable(per[k][2])) for k in fused])
            for k, (gx_k, gw_k) in zip(fused, res):
                gxs[k], gws[k] = gx_k, gw_k
        with conv_group():                                    # the other data gradients: one launch per kernel instantiation
            for k in range(K):
                if k in fused:
                    continue
                x, ldx, weight, wd, g, dy, sums, N, H, W, CinP, CoutP = per[k][:12]
                dil = ctx.meta[k][0]
                KH, KW = weight.shape[2], weight.shape[3]
                if ctx.needs_input_grad[1 + 5 * k]:
                    gxs[k] = torch.empty((N, H, W, CinP), dtype=x.dtype, device=x.device)
                    L.call("egm_conv_fwd", dt, ptr(dy), CoutP, ptr(wd), None, 0, ptr(gxs[k]), CinP, None, N, H, W, CoutP, CinP, KH, KW, dil, st)
        with conv_group():                                    # ... and their weight gradients (slab kernels)
            for k in range(K):
                if k in fused:
                    continue
                x, ldx, weight, wd, g, dy, sums, N, H, W, CinP, CoutP = per[k][:12]
                dil, groups, has_bias, Cin, Cout, act, training = ctx.meta[k]
                if ctx.needs_input_grad[1 + 5 * k + 1]:
                    gws[k] = _conv_wgrad(x, ldx, dy, CoutP, weight, dil, groups, Cin, Cout)
        for k in range(K):
            x, ldx, weight, wd, g, dy, sums, N, H, W, CinP, CoutP = per[k][:12]
            dil, groups, has_bias, Cin, Cout, act, training = ctx.meta[k]
            KH, KW = weight.shape[2], weight.shape[3]
            base = 1 + 5 * k
            gx, gw, gb = gxs[k], gws[k], None
            if has_bias and ctx.needs_input_grad[base + 2]:
                gb = _bn_conv_bias_grad(Cout, dy, training, x.device)
            ggamma = sums[1, :Cout] if ctx.needs_input_grad[base + 3] else None
            gbeta = sums[0, :Cout] if ctx.needs_input_grad[base + 4] else None
            grads += [gx, gw, gb, ggamma, gbeta]
        return tuple(grads)


def multi_conv_bn_act(items):
    """items: K <= 4 tuples (x, conv, bn, act, dil, groups, out) of INDEPENDENT layers (x: NHWC tensor or Lazy) -> list of K outputs.
    With the BatchNorm passes shared between the layers (_MultiConvBN); falls back to K conv_bn_act calls when disabled or K == 1."""
    if not _FUSE_BN_MULTI or len(items) == 1 or len(items) > BN_MULTI_MAX:
        return [conv_bn_act(x, conv, bn, act, dil=dil, groups=groups, out=out) for x, conv, bn, act, dil, groups, out in items]
    meta, flat = [], []
    for x, conv, bn, act, dil, groups, out in items:
        if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_egm_counter_managed", False):
            bn.num_batches_tracked.add_(1)
        training = bn.training or bn.running_mean is None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        meta.append((bn.running_mean, bn.running_var, bn.eps, momentum, act, training, dil, groups, None if out is None else [out]))
        flat += [materialize(x), conv.weight, conv.bias, bn.weight, bn.bias]
    return list(_MultiConvBN.apply(meta, *flat))


def fuse_bn_multi(enabled=None):
    global _FUSE_BN_MULTI
    if enabled is not None:
        _FUSE_BN_MULTI = bool(enabled)
    return _FUSE_BN_MULTI


EW_GATE, EW_SAR = 0, 1                     # enum egm_ew_mode
_FUSE_BN_EW = os.environ.get("EGM_FUSE_BN_EW", "1") != "0"


def fuse_bn_ew(enabled=None):
    """Get / set whether BatchNorm + the element-wise op behind it run as the fused kernels of csrc/bn_fused.hip."""
    global _FUSE_BN_EW
    if enabled is not None:
        _FUSE_BN_EW = bool(enabled)
    return _FUSE_BN_EW


class _ConvBNEw(Function):
    """conv -> BatchNorm(+act) -> element-wise consumer as ONE autograd node (csrc/bn_fused.hip):
         EW_GATE  out = p*(1 + sigmoid(BN(conv(e))))       EdgeAwareFeatureEnhancer   (src/EGM-UNet.py:872-886)
         EW_SAR   out = relu(alpha*p + BN(conv(x_sc)))     EdgeEnhancedGRFB tail      (:1315-1317)
    The BatchNorm output z and its gradient dz never exist in memory: forward reads (y, p) and writes out; backward computes the
    BatchNorm partial sums from (g, p|out, y), then dy and dp in one pass, then the conv's weight and data gradients."""

    @staticmethod
    def forward(ctx, x, p, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, act, training, mode, alpha, out_slot):
        x, ldx = _nhwc(x)
        p, ldp = _nhwc(p)
        Cout, Cin_g = weight.shape[0], weight.shape[1]
        y, stats, wd = _conv_forward(x, ldx, weight, bias, 1, 1, training)
        CoutP, npix, dev = y.shape[3], _npix(y), y.device
        if tuple(p.shape) != tuple(y.shape):
            raise RuntimeError(f"conv_bn_ew: element-wise operand {tuple(p.shape)} does not match the conv output {tuple(y.shape)}")
        L, dt, st = lib(), dtype_code(y.dtype), stream()
        coef = _f32((4, CoutP), dev)
        if training:
            L.call("egm_bn_finalize", ptr(stats), stats.shape[0], npix, ptr(gamma.detach()), ptr(beta.detach()), eps, momentum,
                   ptr(running_mean), ptr(running_var), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), CoutP, Cout, st)
        else:
            L.call("egm_bn_eval_coeffs", ptr(gamma.detach()), ptr(beta.detach()), ptr(running_mean), ptr(running_var), eps,
                   ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), CoutP, Cout, st)
        out, ldo = _slot_or_new(out_slot, tuple(y.shape), y.dtype, dev)
        L.call("egm_bn_ew_fwd", dt, mode, ptr(y), CoutP, ptr(coef[0]), ptr(coef[1]), act, ptr(p), ldp, float(alpha), ptr(out), ldo, npix,
               CoutP, st)
        if ctx.needs_input_grad[2]:
            _note_conv_use(weight)
        # GATE needs p for its backward, SAR the output (ReLU mask)
        ctx.save_for_backward(x, weight, wd, y, coef, p if mode == EW_GATE else out)
        ctx.meta = (bias is not None, Cin_g, Cout, act, training, mode, float(alpha))
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, wd, y, coef, q = ctx.saved_tensors
        has_bias, Cin, Cout, act, training, mode, alpha = ctx.meta
        g, ldg = _nhwc(g)
        x, ldx = _nhwc(x)
        q, ldq = _nhwc(q)
        N, H, W, CinP = x.shape
        CoutP, npix, dev = y.shape[3], _npix(y), y.device
        KH, KW = weight.shape[2], weight.shape[3]
        L, dt, st = lib(), dtype_code(x.dtype), stream()
        nb = L.query("egm_channel_partials_blocks", npix, CoutP)
        part = _f32(nb * 2 * CoutP, dev)
        L.call("egm_bn_ew_bwd_reduce", dt, mode, ptr(g), ldg, ptr(q), ldq, ptr(y), CoutP, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]),
               ptr(coef[3]), act, alpha, ptr(part), npix, CoutP, st)
        sums, cf4 = _f32((2, CoutP), dev), _f32((4, CoutP), dev)
        L.call("egm_bn_bwd_coefs", ptr(part), nb, npix, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), 1 if training else 0,
               ptr(sums), ptr(cf4), CoutP, st)
        dy = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=dev)
        dp = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=dev)
        L.call("egm_bn_ew_bwd_apply", dt, mode, ptr(g), ldg, ptr(q), ldq, ptr(y), CoutP, ptr(cf4), act, alpha, ptr(dy), CoutP, ptr(dp),
               CoutP, npix, CoutP, st)
        gb = None
        gx, gw = _conv_grads(x, ldx, dy, weight, wd, 1, 1, Cin, Cout, ctx.needs_input_grad[0], ctx.needs_input_grad[2])
        if has_bias and ctx.needs_input_grad[3]:
            gb = _bn_conv_bias_grad(Cout, dy, training, dev)
        ggamma = sums[1, :Cout] if ctx.needs_input_grad[4] else None
        gbeta = sums[0, :Cout] if ctx.needs_input_grad[5] else None
        return gx, dp if ctx.needs_input_grad[1] else None, gw, gb, ggamma, gbeta, None, None, None, None, None, None, None, None, None


def conv_bn_ew(x, conv, bn, act, p, mode, alpha=1.0, out=None):
    """F(p, act(BN(conv(x)))) with F = EW_GATE: p*(1+z) or EW_SAR: relu(alpha*p + z); 1x1 convs (the two users in EdgeEnhancedGRFB)."""
    if conv.weight.shape[2] != 1 or conv.weight.shape[3] != 1 or conv.groups != 1:
        raise RuntimeError("conv_bn_ew: 1x1 ungrouped convolutions only")
    if pw_applicable(x, [conv]):
        return pw_conv_bn([(x, [(conv, bn, act, mode, p, alpha, out)])])[0]
    if not _FUSE_BN_EW:
        z = conv_bn_act(x, conv, bn, act)
        if mode == EW_GATE:
            return gate_mul(p, z) if out is None else materialize(gate_mul(p, z), out)
        r = scale_add_relu(p, alpha, z)
        return r if out is None else materialize(r, out)
    if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_egm_counter_managed", False):
        bn.num_batches_tracked.add_(1)
    training = bn.training or bn.running_mean is None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _ConvBNEw.apply(materialize(x), materialize(p), conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                           momentum, act, training, mode, alpha, None if out is None else [out])


# ----------------------------------------------------------------------------------------------------------
# 1x1 conv -> BatchNorm -> activation (-> element-wise consumer) from the input's moments (csrc/pw_bn.hip)
# ----------------------------------------------------------------------------------------------------------
# Built, parity-tested (tests/test_gpu_pw.py) and measured SLOWER than the materialised chain on every site of the headline config
# (+0.55 ms with the conv_bn_ew chains, +0.07 ms with the branch tails): opt-in.
_FUSE_PW = os.environ.get("EGM_PW_BN", "0") != "0"
_PW_HEAD = struct.Struct("<23Qq14i3f4x")          # egm_pw_head (include/egm_hip.h)


def fuse_pw(enabled=None):
    """Get / set whether 1x1 conv -> BatchNorm(+act, + GATE / SAR) chains run in the moment form (no conv output in memory)."""
    global _FUSE_PW
    if enabled is not None:
        _FUSE_PW = bool(enabled)
    return _FUSE_PW


def _pw_head(x=None, w=None, wd=None, bias=None, gamma=None, beta=None, rm=None, rv=None, coef=None, p=None, out=None, g=None, q=None, dp=None,
             sums=None, cf4=None, dw=None, dbias=None, mom=None, cov=None, mu=None, bwd=None, dx=None, npix=0, ldx=0, lddx=0, Cin=0, Cin_real=0,
             Cout=0, CoutP=0, act=0, mode=0, ldp=0, ldo=0, ldg=0, ldq=0, lddp=0, train=0, alpha=1.0, eps=0.0, momentum=0.0):
    dp_ = lambda t: 0 if t is None else t.data_ptr()
    return _PW_HEAD.pack(dp_(x), dp_(w), dp_(wd), dp_(bias), dp_(gamma), dp_(beta), dp_(rm), dp_(rv), dp_(coef), dp_(p), dp_(out), dp_(g), dp_(q),
                         dp_(dp), dp_(sums), dp_(cf4), dp_(dw), dp_(dbias), dp_(mom), dp_(cov), dp_(mu), dp_(bwd), dp_(dx), npix, ldx, lddx, Cin,
                         Cin_real, Cout, CoutP, act, mode, ldp, ldo, ldg, ldq, lddp, train, alpha, eps, momentum)


def _pw_ok(dtype, CinP, couts):
    """couts: padded output channel counts of the heads sharing one input"""
    return bool(lib().cdll.egm_pw_supported(dtype_code(dtype), CinP, sum(couts), len(couts)))


class _PwConvBN(Function):
    """K <= 4 inputs, each feeding one or two 1x1 conv -> BatchNorm -> act (-> GATE / SAR) heads, as ONE autograd node in the moment form
    (csrc/pw_bn.hip): forward = moments + covariance + coefficients + ONE streaming pass per launch; backward = reduce + coefficients
    (with the weight gradients in closed form) + ONE streaming pass that also sums the data gradients of the heads sharing an input.

    meta: list (one entry per input) of lists (one entry per head) of
          (rm, rv, eps, momentum, act, mode, alpha, training, out_slot)            mode: 0 none, 1 GATE, 2 SAR
    flat: per input: x, then per head: weight, bias, gamma, beta, p"""

    @staticmethod
    def forward(ctx, meta, *flat):
        L, st = lib(), stream()
        descs, outs, saved, info, keep = [], [], [], [], []
        pos = 0
        dt = None
        for heads in meta:
            x, ldx = _nhwc(flat[pos])
            xpos = pos
            pos += 1
            N, H, W, CinP = x.shape
            npix, dev = _npix(x), x.device
            dt = dtype_code(x.dtype)
            training_any = any(h[7] for h in heads)
            mom = cov = mu = None
            if training_any:
                mom = _f32(L.query("egm_pw_moments_floats", npix, CinP), dev)
                cov = torch.empty((CinP, CinP), dtype=torch.float64, device=dev)
                mu = torch.empty(CinP, dtype=torch.float64, device=dev)
            hinfo = []
            for (rm, rv, eps, momentum, act, mode, alpha, training, out_slot) in heads:
                weight, bias, gamma, beta, p = flat[pos:pos + 5]
                wpos = pos
                pos += 5
                Cout, Cin = weight.shape[0], weight.shape[1]
                if pad8(Cin) != CinP:
                    raise RuntimeError(f"pw_conv_bn: input has {CinP} channels, weight expects {Cin}")
                CoutP = pad8(Cout)
                wf, wd = _packed_weights(weight, 1, x.dtype)
                coef = _f32((4, CoutP), dev)
                out, ldo = _slot_or_new(out_slot, (N, H, W, CoutP), x.dtype, dev)
                pt, ldp = (None, 0)
                if mode:
                    pt, ldp = _nhwc(p)
                    if tuple(pt.shape) != (N, H, W, CoutP):
                        raise RuntimeError(f"pw_conv_bn: element-wise operand {tuple(pt.shape)} does not match the conv output {(N, H, W, CoutP)}")
                descs.append(_pw_head(x=x, w=wf, wd=wd, bias=bias.detach() if bias is not None else None, gamma=gamma.detach(), beta=beta.detach(),
                                      rm=rm, rv=rv, coef=coef, p=pt, out=out, mom=mom, cov=cov, mu=mu, npix=npix, ldx=ldx, Cin=CinP, Cin_real=Cin,
                                      Cout=Cout, CoutP=CoutP, act=act, mode=mode, ldp=ldp, ldo=ldo, train=1 if training else 0, alpha=float(alpha),
                                      eps=eps, momentum=momentum))
                outs.append(out)
                # GATE needs p for its backward, SAR the output (ReLU mask)
                qsave = pt if mode == 1 else (out if mode == 2 else None)
                hinfo.append((wpos, len(saved), Cout, Cin, CoutP, act, mode, float(alpha), training, bias is not None))
                saved += [weight, wf, wd, coef, qsave]
            info.append((xpos, len(saved), hinfo, training_any))
            saved += [x, cov, mu]
            keep.append(mom)                         # alive until the launches below are enqueued (the allocator would hand it to the next input)
        blob = b"".join(descs)
        if any(i[3] for i in info):
            tr = b"".join(d for d, i in zip(descs, [ii for ii in info for _ in ii[2]]) if i[3])
            L.call("egm_pw_moments", dt, tr, len(tr) // _PW_HEAD.size, st)
        L.call("egm_pw_fwd_coefs", dt, blob, len(descs), st)
        L.call("egm_pw_fwd", dt, blob, len(descs), st)
        ctx.save_for_backward(*[t for t in saved if t is not None])
        ctx.saved_mask = [t is not None for t in saved]
        ctx.info = info
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        L, st = lib(), stream()
        it = iter(ctx.saved_tensors)
        saved = [next(it) if m else None for m in ctx.saved_mask]
        ngrad = max(max(h[0] for h in i[2]) for i in ctx.info) + 5
        grads = [None] * ngrad
        descs, keep = [], []
        k = 0
        dt = None
        for xpos, xs, hinfo, _ in ctx.info:
            x, cov, mu = saved[xs:xs + 3]
            x, ldx = _nhwc(x)
            N, H, W, CinP = x.shape
            npix, dev = _npix(x), x.device
            dt = dtype_code(x.dtype)
            tot = sum(h[4] for h in hinfo)
            bwd = _f32(L.query("egm_pw_bwd_floats", dt, npix, CinP, tot), dev)
            dx = torch.empty((N, H, W, CinP), dtype=x.dtype, device=dev) if ctx.needs_input_grad[1 + xpos] else None
            grads[xpos] = dx
            for (wpos, sp, Cout, Cin, CoutP, act, mode, alpha, training, has_bias) in hinfo:
                weight, wf, wd, coef, qsave = saved[sp:sp + 5]
                g = gouts[k]
                k += 1
                if g is None:
                    g = torch.zeros((N, H, W, CoutP), dtype=x.dtype, device=dev)
                g, ldg = _nhwc(g)
                q, ldq = (None, 0) if qsave is None else _nhwc(qsave)
                sums, cf4 = _f32((2, CoutP), dev), _f32((4, CoutP), dev)
                need_w = ctx.needs_input_grad[1 + wpos]
                dw = torch.empty_like(weight) if need_w else None
                dp = torch.empty((N, H, W, CoutP), dtype=x.dtype, device=dev) if (mode and ctx.needs_input_grad[1 + wpos + 4]) else None
                dbias = None
                if has_bias and ctx.needs_input_grad[1 + wpos + 1]:
                    dbias = _zero_grad_vec(Cout, dev) if training else _f32(CoutP, dev)
                descs.append(_pw_head(x=x, w=wf, wd=wd, coef=coef, g=g, q=q, dp=dp, sums=sums, cf4=cf4, dw=dw, dbias=None if training else dbias,
                                      cov=cov, mu=mu, bwd=bwd, dx=dx, npix=npix, ldx=ldx, lddx=CinP, Cin=CinP, Cin_real=Cin, Cout=Cout, CoutP=CoutP,
                                      act=act, mode=mode, ldg=ldg, ldq=ldq, lddp=CoutP, train=1 if training else 0, alpha=alpha))
                keep += [g, q, sums, cf4, bwd]
                grads[wpos] = dw
                grads[wpos + 1] = dbias if (dbias is None or training) else dbias[:Cout]
                grads[wpos + 2] = sums[1, :Cout] if ctx.needs_input_grad[1 + wpos + 2] else None
                grads[wpos + 3] = sums[0, :Cout] if ctx.needs_input_grad[1 + wpos + 3] else None
                grads[wpos + 4] = dp
        blob = b"".join(descs)
        L.call("egm_pw_bwd_reduce", dt, blob, len(descs), st)
        L.call("egm_pw_bwd_coefs", dt, blob, len(descs), st)
        L.call("egm_pw_bwd_apply", dt, blob, len(descs), st)
        return (None,) + tuple(grads)


def pw_conv_bn(problems):
    """problems: list (<= 4) of (x, heads), heads = list (<= 2) of (conv, bn, act, mode, p, alpha, out) with mode None / EW_GATE / EW_SAR:
    out_h = F(p, act(BN(conv1x1_h(x)))) for every head, computed in the moment form.  -> flat list of outputs (problem-major).
    Callers check pw_applicable() first."""
    meta, flat = [], []
    for x, heads in problems:
        flat.append(materialize(x))
        hm = []
        for conv, bn, act, mode, p, alpha, out in heads:
            if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_egm_counter_managed", False):
                bn.num_batches_tracked.add_(1)
            training = bn.training or bn.running_mean is None
            momentum = 0.1 if bn.momentum is None else bn.momentum
            hm.append((bn.running_mean, bn.running_var, bn.eps, momentum, act, 0 if mode is None else mode + 1, alpha, training,
                       None if out is None else [out]))
            flat += [conv.weight, conv.bias, bn.weight, bn.bias, None if mode is None else materialize(p)]
        meta.append(hm)
    return list(_PwConvBN.apply(meta, *flat))


_PW_SITES = set(t for t in os.environ.get("EGM_PW_SITES", "ew,heads,tails").split(",") if t)


def pw_sites(sites=None):
    """Get / set the places of EdgeEnhancedGRFB that use the moment form: "ew" (the conv_bn_ew chains: both EdgeAwareFeatureEnhancers and
    the shortcut), "heads" (the two 1x1 branch heads sharing the enhanced input), "tails" (the three 1x1 branch tails)."""
    global _PW_SITES
    if sites is not None:
        _PW_SITES = set(sites)
    return set(_PW_SITES)


def pw_applicable(x, convs, site="ew"):
    """True when the 1x1 convs `convs` (nn.Conv2d holders) reading the NHWC tensor / Lazy x can take the moment form."""
    if not _FUSE_PW or site not in _PW_SITES:
        return False
    for c in convs:
        if tuple(c.weight.shape[2:]) != (1, 1) or c.groups != 1 or pad8(c.weight.shape[1]) != x.shape[3]:
            return False
    return _pw_ok(x.dtype, x.shape[3], [pad8(c.weight.shape[0]) for c in convs])


def _slot_or_new(out_slot, shape, dtype, device):
    """Output placement: `out_slot` is None or a one-element list holding a kernel-addressable NHWC view (a channel slice of a wider
    buffer, e.g. of a concat destination) the result is written into -- the concat copy and its extra tensor write disappear.  The
    list keeps the tensor out of autograd's sight: it is returned as a fresh output, not as an input passed through."""
    if out_slot is None:
        t = torch.empty(shape, dtype=dtype, device=device)
        return t, shape[3]
    t = out_slot[0]
    v, ld = _nhwc(t)
    if v.data_ptr() != t.data_ptr() or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
        raise RuntimeError(f"egm_unet_amd: output slot {tuple(t.shape)}/{t.dtype} does not fit result {tuple(shape)}/{dtype}")
    return t, ld


def cat_slots(N, H, W, channels, dtype, device):
    """A concat destination [N, H, W, sum(channels)] and its channel-slice views, for producers that write in place."""
    buf = torch.empty((N, H, W, sum(channels)), dtype=dtype, device=device)
    views, off = [], 0
    for c in channels:
        views.append(buf[..., off:off + c])
        off += c
    return buf, views


def bn_act(y, bn, act, stats=None, out=None):
    """bn: an nn.BatchNorm2d used as the parameter/buffer holder."""
    training = bn.training or bn.running_mean is None
    if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_egm_counter_managed", False):
        bn.num_batches_tracked.add_(1)                  # bookkeeping counter (int64), as nn.BatchNorm2d does
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _BnAct.apply(y, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, act, training,
                        None if out is None else [out])


def conv_bn_act(x, conv, bn, act, dil=1, groups=1, out=None, lazy=False):
    """conv -> BatchNorm -> activation (statistics from the conv epilogue).  x: NHWC tensor (a Lazy is materialised first).
    lazy="force" returns the result as a Lazy for a consumer that applies the BatchNorm in a pass of its own (MCALayer statistics, the
    classifier inside the apply pass); otherwise the result is materialised, into the destination view `out` when given."""
    if bn.training and bn.num_batches_tracked is not None and not getattr(bn, "_egm_counter_managed", False):
        bn.num_batches_tracked.add_(1)                  # bookkeeping counter (int64), as nn.BatchNorm2d does
    training = bn.training or bn.running_mean is None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    x = materialize(x)
    y, coef = _ConvBN.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, act,
                            training, dil, groups, getattr(x, "_egm_split", 0))
    z = Lazy(y, coef, act)
    if out is None and lazy == "force":                 # the consumer materialises the tensor in a pass of its own
        return z
    return z.materialize(out)


# ----------------------------------------------------------------------------------------------------------
# pooling / upsample + concat
# ----------------------------------------------------------------------------------------------------------
class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x):
        x, ldx = _nhwc(x)
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        lib().call("egm_maxpool2_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(y), C, N, H, W, C, stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        x, ldx = _nhwc(x)
        gy, ldg = _nhwc(gy)
        N, H, W, C = x.shape
        gx = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
        lib().call("egm_maxpool2_bwd", dtype_code(x.dtype), ptr(x), ldx, ptr(gy), ldg, ptr(gx), C, N, H, W, C, stream())
        return gx


def maxpool2(x):
    return _MaxPool2.apply(x)


class _ForkMaxPool2(Function):
    """x -> (alias of x for the skip connection, maxpool2(x)): one autograd node, so that backward adds the skip gradient while it
    scatters the pooled one (egm_maxpool2_bwd_add) instead of a scatter pass followed by an axpby pass."""

    @staticmethod
    def forward(ctx, x):
        xk, ldx = _nhwc(x)
        N, H, W, C = xk.shape
        y = torch.empty((N, H // 2, W // 2, C), dtype=xk.dtype, device=xk.device)
        lib().call("egm_maxpool2_fwd", dtype_code(xk.dtype), ptr(xk), ldx, ptr(y), C, N, H, W, C, stream())
        ctx.save_for_backward(xk)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, gskip, gy):
        (x,) = ctx.saved_tensors
        x, ldx = _nhwc(x)
        N, H, W, C = x.shape
        if gy is None:
            return gskip
        gy, ldg = _nhwc(gy)
        gx = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
        dt = dtype_code(x.dtype)
        if gskip is None:
            lib().call("egm_maxpool2_bwd", dt, ptr(x), ldx, ptr(gy), ldg, ptr(gx), C, N, H, W, C, stream())
        elif (H | W) & 1:                                       # odd sizes: the zero-tail pass, then the sum
            lib().call("egm_maxpool2_bwd", dt, ptr(x), ldx, ptr(gy), ldg, ptr(gx), C, N, H, W, C, stream())
            gx = _axpby(gx, 1.0, gskip, 1.0)
        else:
            gs, lds = _nhwc(gskip)
            lib().call("egm_maxpool2_bwd_add", dt, ptr(x), ldx, ptr(gy), ldg, ptr(gs), lds, ptr(gx), C, N, H, W, C, stream())
        return gx


def fork_maxpool2(x):
    """(x for the skip connection, maxpool2(x)) with the two gradients of x summed inside the pooling backward."""
    return _ForkMaxPool2.apply(x)


class _UpCat(Function):
    """cat([skip, pad(bilinear_x2(low))], channel)"""

    @staticmethod
    def forward(ctx, skip, low, catbuf=None):
        skip, lds = _nhwc(skip)
        low, ldl = _nhwc(low)
        N, Hs, Ws, Cs = skip.shape
        _, Hl, Wl, Cl = low.shape
        inplace = False
        if catbuf is not None:
            # the skip tensor was produced straight into the first Cs channels of the concat destination: only the upsampled half is
            # written (a tensor write costs about twice a read on this part, and the skip copy was two thirds of this kernel's bytes)
            cb = catbuf[0]
            inplace = (cb.is_contiguous() and tuple(cb.shape) == (N, Hs, Ws, Cs + Cl) and cb.dtype == skip.dtype
                       and cb.data_ptr() == skip.data_ptr() and lds == Cs + Cl)
        if inplace:
            out = cb
            lib().call("egm_upcat_fwd", dtype_code(skip.dtype), None, lds, ptr(low), ldl, ptr(out), Cs + Cl, N, Hs, Ws, Cs,
                       Hl, Wl, Cl, stream())
        else:
            out = torch.empty((N, Hs, Ws, Cs + Cl), dtype=skip.dtype, device=skip.device)
            lib().call("egm_upcat_fwd", dtype_code(skip.dtype), ptr(skip), lds, ptr(low), ldl, ptr(out), Cs + Cl, N, Hs, Ws, Cs,
                       Hl, Wl, Cl, stream())
        ctx.shape = (N, Hs, Ws, Cs, Hl, Wl, Cl)
        return out

    @staticmethod
    def backward(ctx, g):
        N, Hs, Ws, Cs, Hl, Wl, Cl = ctx.shape
        pend = _DEFERRED_DZ.pop(g.data_ptr(), None) if _DEFERRED_DZ else None
        gskip = glow = None
        if pend is not None:                           # the conv behind the concat wrote the two halves of its gradient as dense tensors
            _, ga, gb = pend[1]
            if ctx.needs_input_grad[0]:
                gskip = ga
            if ctx.needs_input_grad[1]:
                glow = torch.empty((N, Hl, Wl, Cl), dtype=gb.dtype, device=gb.device)
                lib().call("egm_upcat_bwd_low", dtype_code(gb.dtype), ptr(gb), Cl, ptr(glow), Cl, N, Hs, Ws, 0, Hl, Wl, Cl, stream())
            return gskip, glow, None
        g, ldo = _nhwc(g)
        if ctx.needs_input_grad[0]:
            gskip = g[..., :Cs]                        # a view: consumers take (ptr, ld)
        if ctx.needs_input_grad[1]:
            glow = torch.empty((N, Hl, Wl, Cl), dtype=g.dtype, device=g.device)
            lib().call("egm_upcat_bwd_low", dtype_code(g.dtype), ptr(g), ldo, ptr(glow), Cl, N, Hs, Ws, Cs, Hl, Wl, Cl, stream())
        return gskip, glow, None


def upcat(skip, low, catbuf=None):
    """catbuf: the concat destination whose first channels ARE `skip` (see cat_slots), or None for a fresh tensor + copy."""
    out = _UpCat.apply(skip, low, None if catbuf is None else [catbuf])
    # a hint for the conv that consumes the concatenation (its only consumer in Up.forward, src/EGM-UNet.py:947-949): the gradient may come
    # back as two dense tensors, split at this channel (ops._conv_dgrad).  Lost -- harmlessly -- if the tensor passes through anything else.
    out._egm_split = skip.shape[3]
    return out


class _Fork2(Function):
    """A tensor consumed twice: the two gradients are summed by the HIP axpby kernel (not by autograd's add)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None:
            return gb
        if gb is None:
            return ga
        ga, lda = _nhwc(ga)
        gb, ldb = _nhwc(gb)
        out = torch.empty(ga.shape, dtype=ga.dtype, device=ga.device)
        lib().call("egm_axpby", dtype_code(ga.dtype), ptr(ga), lda, 1.0, ptr(gb), ldb, 1.0, ptr(out), out.shape[3], _npix(ga),
                   ga.shape[3], stream())
        return out


def fork2(x):
    return _Fork2.apply(x)


# ----------------------------------------------------------------------------------------------------------
# fan-out / channel split / concat (gradient plumbing done by the HIP axpby kernel, never by autograd's add)
# ----------------------------------------------------------------------------------------------------------
def _axpby(a, alpha, b, beta, out=None):
    a, lda = _nhwc(a)
    ldb = 0
    if b is not None:
        b, ldb = _nhwc(b)
    if out is None:
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    o, ldo = _nhwc(out)
    assert o.data_ptr() == out.data_ptr(), "axpby: output view must be kernel-addressable"
    lib().call("egm_axpby", dtype_code(a.dtype), ptr(a), lda, float(alpha), ptr(b), ldb, float(beta), ptr(o), ldo, _npix(a),
               a.shape[3], stream())
    return out


class _Fork(Function):
    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        rest = gs[1:]
        while rest:
            if len(rest) >= 2:                          # 3-4 gradients summed in one pass
                take, rest = rest[:3], rest[3:]
                ts = [_nhwc(t) for t in [acc] + take]
                out = torch.empty(ts[0][0].shape, dtype=ts[0][0].dtype, device=ts[0][0].device)
                d = ts[3] if len(ts) > 3 else (None, 0)
                lib().call("egm_sum4", dtype_code(out.dtype), ptr(ts[0][0]), ts[0][1], ptr(ts[1][0]), ts[1][1], ptr(ts[2][0]), ts[2][1],
                           ptr(d[0]), d[1], ptr(out), out.shape[3], _npix(out), out.shape[3], stream())
                acc = out
            else:
                acc = _axpby(acc, 1.0, rest[0], 1.0)
                rest = rest[1:]
        return acc, None


def fork(x, n):
    """n autograd aliases of x whose gradients are summed in one pass; a Lazy forks into n Lazies over aliases of its stand-in."""
    if isinstance(x, Lazy):
        return tuple(Lazy(y, x.coef, x.act) for y in _Fork.apply(x.y, n))
    return _Fork.apply(x, n)


class _SplitC(Function):
    """x[..., :c0], x[..., c0:] as views (c0 multiple of 8); backward re-assembles one dense gradient."""

    @staticmethod
    def forward(ctx, x, c0):
        ctx.c0 = c0
        return x[..., :c0], x[..., c0:]

    @staticmethod
    def backward(ctx, ga, gb):
        c0 = ctx.c0
        ref = ga if ga is not None else gb
        N, H, W = ref.shape[:3]
        Ca = c0
        Cb = (gb.shape[3] if gb is not None else 0)
        if ga is None or gb is None:
            raise RuntimeError("split_channels: both halves must receive a gradient")
        out = torch.empty((N, H, W, Ca + Cb), dtype=ref.dtype, device=ref.device)
        _axpby(ga, 1.0, None, 0.0, out[..., :Ca])
        _axpby(gb, 1.0, None, 0.0, out[..., Ca:])
        return out, None


def split_channels(x, c0):
    return _SplitC.apply(x, c0)


class _CatC(Function):
    @staticmethod
    def forward(ctx, buf_slot, *xs):
        N, H, W = xs[0].shape[:3]
        cs = [x.shape[3] for x in xs]
        buf = buf_slot[0] if buf_slot is not None else None
        if buf is not None and not (buf.is_contiguous() and tuple(buf.shape) == (N, H, W, sum(cs)) and buf.dtype == xs[0].dtype):
            raise RuntimeError("cat_channels: destination buffer does not match the inputs")
        out = buf if buf is not None else torch.empty((N, H, W, sum(cs)), dtype=xs[0].dtype, device=xs[0].device)
        off = 0
        for x, c in zip(xs, cs):
            dst = out[..., off:off + c]
            # an input that was produced straight into its slot (bn_act / gate3 with out=) needs no copy
            if not (x.data_ptr() == dst.data_ptr() and x.stride() == dst.stride()):
                _axpby(x, 1.0, None, 0.0, dst)
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, g):
        outs, off = [None], 0
        for c in ctx.cs:
            outs.append(g[..., off:off + c])
            off += c
        return tuple(outs)


def cat_channels(xs, buf=None):
    """buf: optional destination from cat_slots(); inputs already living in their slot are not copied."""
    return _CatC.apply(None if buf is None else [buf], *xs)


# ----------------------------------------------------------------------------------------------------------
# elementwise pieces of the EGM blocks
# ----------------------------------------------------------------------------------------------------------
def _hp(x):
    x, ldx = _nhwc(x)
    N, H, W, C = x.shape
    out = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    lib().call("egm_highpass3", dtype_code(x.dtype), ptr(x), ldx, ptr(out), C, N, H, W, C, stream())
    return out


class _Highpass3(Function):
    @staticmethod
    def forward(ctx, x):
        return _hp(x)

    @staticmethod
    def backward(ctx, g):
        return _hp(g)           # x - avg3(x) with zero padding is self-adjoint


def highpass3(x):
    return _Highpass3.apply(x)


class _ForkHighpass(Function):
    """x -> (highpass3(x), n aliases of x) as one node: backward computes highpass3(g_hp) + sum of the other gradients in ONE pass
    (egm_sum4_hp) instead of a stencil pass that writes its result and a fan-in pass that reads it back."""

    @staticmethod
    def forward(ctx, x, n):
        return (_hp(x),) + tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, ghp, *gs):
        gs = [g for g in gs if g is not None]
        if ghp is None:
            return (_Fork.backward(ctx, *gs)[0] if gs else None), None
        if not gs:
            return _hp(ghp), None
        while len(gs) > 3:                                    # more than three other consumers: pre-sum the tail
            gs = gs[:2] + [_Fork.backward(ctx, *gs[2:])[0]]
        a, lda = _nhwc(ghp)
        N, H, W, C = a.shape
        ts = [_nhwc(t) for t in gs] + [(None, 0)] * (3 - len(gs))
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
        lib().call("egm_sum4_hp", dtype_code(a.dtype), ptr(a), lda, ptr(ts[0][0]), ts[0][1], ptr(ts[1][0]), ts[1][1], ptr(ts[2][0]), ts[2][1],
                   ptr(out), C, N, H, W, C, stream())
        return out, None


def fork_highpass3(x, n):
    """(highpass3(x), alias_1, ..., alias_n): the edge extractor of EdgeAwareFeatureEnhancer together with x's other consumers."""
    return _ForkHighpass.apply(x, n)


class _GateMul(Function):
    @staticmethod
    def forward(ctx, x, w):
        x, ldx = _nhwc(x)
        w, ldw = _nhwc(w)
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("egm_gate_mul_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(w), ldw, ptr(out), x.shape[3], _npix(x), x.shape[3], stream())
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        x, ldx = _nhwc(x); w, ldw = _nhwc(w); g, ldg = _nhwc(g)
        C = x.shape[3]
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        dw = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("egm_gate_mul_bwd", dtype_code(x.dtype), ptr(g), ldg, ptr(x), ldx, ptr(w), ldw, ptr(dx), C, ptr(dw), C, _npix(x), C, stream())
        return dx, dw


def gate_mul(x, w):
    return _GateMul.apply(x, w)


class _ScaleAddRelu(Function):
    @staticmethod
    def forward(ctx, a, alpha, b):
        a, lda = _nhwc(a); b, ldb = _nhwc(b)
        C = a.shape[3]
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
        lib().call("egm_scale_add_relu_fwd", dtype_code(a.dtype), ptr(a), lda, float(alpha), ptr(b), ldb, ptr(out), C, _npix(a), C, stream())
        ctx.save_for_backward(out)
        ctx.alpha = float(alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        out, ldo = _nhwc(out); g, ldg = _nhwc(g)
        C = out.shape[3]
        da = torch.empty(out.shape, dtype=out.dtype, device=out.device)
        db = torch.empty(out.shape, dtype=out.dtype, device=out.device)
        lib().call("egm_scale_add_relu_bwd", dtype_code(out.dtype), ptr(g), ldg, ptr(out), ldo, ctx.alpha, ptr(da), C, ptr(db), C, _npix(out), C,
                   stream())
        return da, None, db


def scale_add_relu(a, alpha, b):
    return _ScaleAddRelu.apply(a, alpha, b)


class _Gate3(Function):
    @staticmethod
    def forward(ctx, x, t, out_slot=None):
        x, ldx = _nhwc(x); t, ldt = _nhwc(t)
        C = x.shape[3]
        out, ldo = _slot_or_new(out_slot, tuple(x.shape), x.dtype, x.device)
        lib().call("egm_gate3_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(t), ldt, ptr(out), ldo, _npix(x), C, stream())
        ctx.save_for_backward(x, t)
        return out

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        x, ldx = _nhwc(x); t, ldt = _nhwc(t); g, ldg = _nhwc(g)
        C = x.shape[3]
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        dt = torch.empty(t.shape[:3] + (8,), dtype=x.dtype, device=x.device)
        lib().call("egm_gate3_bwd", dtype_code(x.dtype), ptr(g), ldg, ptr(x), ldx, ptr(t), ldt, ptr(dx), C, ptr(dt), 8, _npix(x), C, stream())
        return dx, dt, None


def gate3(x, t, out=None):
    return _Gate3.apply(x, t, None if out is None else [out])


class _Gate3Pool(Function):
    """gate3 followed by the skip-connection fork + max pool as ONE node: -> (out, maxpool2(out)); backward takes both gradients and
    never materialises their sum (egm_gate3_pool_bwd; EdgeEnhancedGRFB's last op, src/EGM-UNet.py:1319-1321, behind :908)."""

    @staticmethod
    def forward(ctx, x, t, out_slot=None):
        x, ldx = _nhwc(x); t, ldt = _nhwc(t)
        N, H, W, C = x.shape
        out, ldo = _slot_or_new(out_slot, tuple(x.shape), x.dtype, x.device)
        pooled = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        lib().call("egm_gate3_fwd_pool", dtype_code(x.dtype), ptr(x), ldx, ptr(t), ldt, ptr(out), ldo, ptr(pooled), C, N, H, W, C, stream())
        ctx.save_for_backward(x, t, out)
        ctx.set_materialize_grads(False)
        return out, pooled

    @staticmethod
    def backward(ctx, gskip, gpool):
        if gskip is None and gpool is None:
            return None, None, None
        x, t, out = ctx.saved_tensors
        x, ldx = _nhwc(x); t, ldt = _nhwc(t)
        N, H, W, C = x.shape
        L, dt_, st = lib(), dtype_code(x.dtype), stream()
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        dt = torch.empty(t.shape[:3] + (8,), dtype=x.dtype, device=x.device)
        if gskip is not None and gpool is not None:
            gs, lds = _nhwc(gskip); gp, ldp = _nhwc(gpool)
            L.call("egm_gate3_pool_bwd", dt_, ptr(gs), lds, ptr(gp), ldp, ptr(x), ldx, ptr(t), ldt, ptr(dx), C, ptr(dt), 8, N, H, W, C, st)
            return dx, dt, None
        if gskip is None:                                       # only the pooled branch was used: the separate scatter
            ok, ldo = _nhwc(out); gp, ldp = _nhwc(gpool)
            gskip = torch.empty(x.shape, dtype=x.dtype, device=x.device)
            L.call("egm_maxpool2_bwd", dt_, ptr(ok), ldo, ptr(gp), ldp, ptr(gskip), C, N, H, W, C, st)
        g, ldg = _nhwc(gskip)
        L.call("egm_gate3_bwd", dt_, ptr(g), ldg, ptr(x), ldx, ptr(t), ldt, ptr(dx), C, ptr(dt), 8, _npix(x), C, st)
        return dx, dt, None


def gate3_pool(x, t, out=None):
    """-> (gate3(x, t) [into `out`], maxpool2 of it); see _Gate3Pool.  Callers check pool_fusable(x) first."""
    return _Gate3Pool.apply(x, t, None if out is None else [out])


class _BcastGate(Function):
    @staticmethod
    def forward(ctx, a, gl):
        a, lda = _nhwc(a); gl, ldg = _nhwc(gl)
        C = a.shape[3]
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
        lib().call("egm_bcast_gate_fwd", dtype_code(a.dtype), ptr(a), lda, ptr(gl), ldg, ptr(out), C, _npix(a), C, stream())
        ctx.save_for_backward(a, gl)
        return out

    @staticmethod
    def backward(ctx, g):
        a, gl = ctx.saved_tensors
        a, lda = _nhwc(a); gl, ldgl = _nhwc(gl); g, ldg = _nhwc(g)
        C = a.shape[3]
        da = torch.empty(a.shape, dtype=a.dtype, device=a.device)
        dgl = torch.empty(gl.shape[:3] + (8,), dtype=a.dtype, device=a.device)
        lib().call("egm_bcast_gate_bwd", dtype_code(a.dtype), ptr(g), ldg, ptr(a), lda, ptr(gl), ldgl, ptr(da), C, ptr(dgl), 8, _npix(a), C,
                   stream())
        return da, dgl


def bcast_gate(a, gl):
    return _BcastGate.apply(a, gl)


class _Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        x, ldx = _nhwc(x)
        C = x.shape[3]
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("egm_gelu_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(out), C, _npix(x), C, stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        x, ldx = _nhwc(x); g, ldg = _nhwc(g)
        C = x.shape[3]
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("egm_gelu_bwd", dtype_code(x.dtype), ptr(g), ldg, ptr(x), ldx, ptr(dx), C, _npix(x), C, stream())
        return dx


def gelu(x):
    return _Gelu.apply(x)


class _Act(Function):
    """Standalone activation (the ReLU inside ChannelAttentionModule.fc) through the BN-apply kernels with unit scale."""

    @staticmethod
    def forward(ctx, x, act):
        x, ldx = _nhwc(x)
        C = x.shape[3]
        coef = torch.zeros((4, C), dtype=torch.float32, device=x.device)
        lib().call("egm_fill_f32", ptr(coef[0]), 1.0, C, stream())
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("egm_bn_act_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(coef[0]), ptr(coef[1]), act, ptr(out), C, _npix(x), C, stream())
        ctx.save_for_backward(x, coef)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, g):
        x, coef = ctx.saved_tensors
        x, ldx = _nhwc(x); g, ldg = _nhwc(g)
        C = x.shape[3]
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("egm_bn_act_bwd_apply", dtype_code(x.dtype), ptr(g), ldg, ptr(x), ldx, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]),
                   ptr(coef[3]), ctx.act, 0, ptr(coef[2]), ptr(dx), C, _npix(x), C, stream())
        return dx, None


def act(x, code):
    return _Act.apply(x, code)


class _ChanMeanMax(Function):
    @staticmethod
    def forward(ctx, x, c_real):
        x, ldx = _nhwc(x)
        out = torch.empty(x.shape[:3] + (8,), dtype=x.dtype, device=x.device)
        lib().call("egm_chan_meanmax_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(out), 8, _npix(x), x.shape[3], c_real, stream())
        ctx.save_for_backward(x)
        ctx.c_real = c_real
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        x, ldx = _nhwc(x); g, ldg = _nhwc(g)
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        lib().call("egm_chan_meanmax_bwd", dtype_code(x.dtype), ptr(g), ldg, ptr(x), ldx, ptr(dx), x.shape[3], _npix(x), x.shape[3], ctx.c_real,
                   stream())
        return dx, None


def chan_meanmax(x, c_real):
    return _ChanMeanMax.apply(x, c_real)


class _GlobalAvgMax(Function):
    """-> [2N, 1, 1, C]: rows 0..N-1 adaptive_avg_pool2d(x, 1), rows N..2N-1 adaptive_max_pool2d(x, 1)."""

    @staticmethod
    def forward(ctx, x):
        x, ldx = _nhwc(x)
        N, H, W, C = x.shape
        L = lib()
        out = torch.empty((2 * N, 1, 1, C), dtype=x.dtype, device=x.device)
        argidx = torch.empty((N, C), dtype=torch.int32, device=x.device)
        ws = torch.empty(L.query("egm_global_pool_workspace", N, H * W, C) // 4 + 4, dtype=torch.float32, device=x.device)
        L.call("egm_global_avgmax_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(out), ptr(argidx), ptr(ws), N, H * W, C, stream())
        ctx.save_for_backward(argidx)
        ctx.shape = (N, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (argidx,) = ctx.saved_tensors
        N, H, W, C = ctx.shape
        g = g.contiguous()
        dx = torch.empty((N, H, W, C), dtype=g.dtype, device=g.device)
        lib().call("egm_global_avgmax_bwd", dtype_code(g.dtype), ptr(g), ptr(argidx), ptr(dx), C, N, H * W, C, stream())
        return dx


def global_avgmax(x):
    return _GlobalAvgMax.apply(x)


class _CaMlp(Function):
    """ChannelAttentionModule.fc on the pooled rows [R,1,1,C]: W2 . relu(W0 . p) as one launch forward, one backward (egm_ca_mlp_*)."""

    @staticmethod
    def forward(ctx, pooled, w0, w2):
        R, Cp = pooled.shape[0], pooled.shape[3]
        Cr, C = w0.shape[0], w0.shape[1]
        pooled = pooled.contiguous()
        w0c, w2c = w0.detach().contiguous(), w2.detach().contiguous()
        h = torch.empty((R, Cr), dtype=torch.float32, device=pooled.device)
        out = (torch.empty if Cp == C else torch.zeros)((R, 1, 1, Cp), dtype=pooled.dtype, device=pooled.device)
        lib().call("egm_ca_mlp_fwd", dtype_code(pooled.dtype), ptr(pooled), Cp, ptr(w0c), ptr(w2c), ptr(h), ptr(out), Cp, R, C, Cr, stream())
        ctx.save_for_backward(pooled, h, w0c, w2c)
        return out

    @staticmethod
    def backward(ctx, g):
        pooled, h, w0, w2 = ctx.saved_tensors
        R, Cp = pooled.shape[0], pooled.shape[3]
        Cr, C = w0.shape[0], w0.shape[1]
        g = g.contiguous()
        dw0, dw2 = torch.empty_like(w0), torch.empty_like(w2)
        dp = (torch.empty if Cp == C else torch.zeros)(pooled.shape, dtype=pooled.dtype, device=pooled.device)
        lib().call("egm_ca_mlp_bwd", dtype_code(pooled.dtype), ptr(g), Cp, ptr(pooled), Cp, ptr(h), ptr(w0), ptr(w2), ptr(dw0), ptr(dw2),
                   ptr(dp), Cp, R, C, Cr, stream())
        return dp, dw0, dw2


def ca_mlp(pooled, w0, w2):
    return _CaMlp.apply(pooled, w0, w2)


class _FusionCombine(Function):
    """f + s*sigmoid(sa[...,0])*sigmoid(ca_avg + ca_max); ca: [2N,1,1,C]."""

    @staticmethod
    def forward(ctx, f, s, sa, ca):
        f, ldf = _nhwc(f); s, lds = _nhwc(s); sa, ldsa = _nhwc(sa)
        ca = ca.contiguous()
        N, H, W, C = f.shape
        out = torch.empty(f.shape, dtype=f.dtype, device=f.device)
        lib().call("egm_fusion_combine_fwd", dtype_code(f.dtype), ptr(f), ldf, ptr(s), lds, ptr(sa), ldsa, ptr(ca), ptr(out), C, N, H * W, C,
                   stream())
        ctx.save_for_backward(s, sa, ca)
        return out

    @staticmethod
    def backward(ctx, g):
        s, sa, ca = ctx.saved_tensors
        s, lds = _nhwc(s); sa, ldsa = _nhwc(sa); g, ldg = _nhwc(g)
        N, H, W, C = s.shape
        L, dev = lib(), s.device
        ds = torch.empty(s.shape, dtype=s.dtype, device=dev)
        dsa = torch.empty((N, H, W, 8), dtype=s.dtype, device=dev)
        nb = L.query("egm_fusion_combine_blocks", H * W, C)
        part = _f32((N, nb, 2, C), dev)
        L.call("egm_fusion_combine_bwd", dtype_code(s.dtype), ptr(g), ldg, ptr(s), lds, ptr(sa), ldsa, ptr(ca), ptr(ds), C, ptr(dsa), 8,
               ptr(part), N, H * W, C, stream())
        red = _f32((N, 2, C), dev)
        L.call("egm_reduce_tiles_batched", ptr(part), N, nb, C, ptr(red), stream())
        # d(ca_avg) = d(ca_max) = dca  ->  [2N,1,1,C] in the activation dtype
        dca = torch.empty((2 * N, 1, 1, C), dtype=s.dtype, device=dev)
        _f32_rows_to_act(red, dca, N, C)
        return g, ds, dsa, dca


def _f32_rows_to_act(red, dca, N, C):
    """red fp32 [N][2][C] (row 0 = dca) -> dca[n] and dca[N+n] in the activation dtype (one launch)."""
    lib().call("egm_rows_dup", dtype_code(dca.dtype), ptr(red), ptr(dca), C, N, C, stream())


def fusion_combine(f, s, sa, ca):
    return _FusionCombine.apply(f, s, sa, ca)


def _register_pack(weight, dtype, wf, wd):
    """Operand packs produced together with a derived weight (fold2 / merge357): what _packed_weights would build for it."""
    key = (weight.data_ptr(), weight._version, _weight_generation[0], dtype, 1, tuple(weight.shape))
    _pack_cache[id(weight)] = (key, wf, wd, weakref.ref(weight))


class _Fold2(Function):
    @staticmethod
    def forward(ctx, w, pack_dtype):                       # [rows, 2K, 1, 1] -> [rows, K, 1, 1]
        rows, K2 = w.shape[0], w.shape[1]
        K = K2 // 2
        out = torch.empty((rows, K, 1, 1), dtype=torch.float32, device=w.device)
        if pack_dtype is None:
            lib().call("egm_fold2_fwd", ptr(w.detach().contiguous()), ptr(out), rows, K, stream())
            _Fold2.packs = None
        else:                                              # the conv operand packs of the folded weight from the same launch
            wf = torch.empty((1, pad8(rows), pad8(K)), dtype=pack_dtype, device=w.device)
            wd = torch.empty((1, pad8(K), pad8(rows)), dtype=pack_dtype, device=w.device)
            lib().call("egm_fold2_pack", dtype_code(pack_dtype), ptr(w.detach().contiguous()), ptr(out), ptr(wf), ptr(wd), rows, K, stream())
            _Fold2.packs = (wf, wd)
        return out

    @staticmethod
    def backward(ctx, g):
        rows, K = g.shape[0], g.shape[1]
        dw = torch.empty((rows, 2 * K, 1, 1), dtype=torch.float32, device=g.device)
        lib().call("egm_fold2_bwd", ptr(g.contiguous()), ptr(dw), rows, K, stream())
        return dw, None


def fold2(w, pack_dtype=None):
    """w[:, :K] + w[:, K:]; pack_dtype: also build the conv operand packs of the result for that activation dtype (one launch)."""
    out = _Fold2.apply(w, pack_dtype)
    if pack_dtype is not None and _Fold2.packs is not None:
        _register_pack(out, pack_dtype, *_Fold2.packs)
        _Fold2.packs = None
    return out


class _SpreadCols(Function):
    """[rows, sum(real), 1, 1] -> [rows, sum(pad8(real)), 1, 1]: the input-channel columns of a 1x1 weight moved to where the channels sit
    after each source tensor was padded to a multiple of 8 on its own (zeros in the gaps).  Backward gathers the columns back."""

    @staticmethod
    def forward(ctx, w, real):
        rows, st = w.shape[0], stream()
        tot_r, tot_p = sum(real), sum(pad8(c) for c in real)
        out = torch.zeros((rows, tot_p, 1, 1), dtype=torch.float32, device=w.device)
        wc = w.detach().contiguous()
        o_r = o_p = 0
        for c in real:
            lib().call("egm_copy_cols_f32", ptr(wc), tot_r, o_r, ptr(out), tot_p, o_p, rows, c, st)
            o_r, o_p = o_r + c, o_p + pad8(c)
        ctx.real = tuple(real)
        return out

    @staticmethod
    def backward(ctx, g):
        real = ctx.real
        rows, st = g.shape[0], stream()
        tot_r, tot_p = sum(real), sum(pad8(c) for c in real)
        gc = g.contiguous()
        dw = torch.empty((rows, tot_r, 1, 1), dtype=torch.float32, device=g.device)
        o_r = o_p = 0
        for c in real:
            lib().call("egm_copy_cols_f32", ptr(gc), tot_p, o_p, ptr(dw), tot_r, o_r, rows, c, st)
            o_r, o_p = o_r + c, o_p + pad8(c)
        return dw, None


def spread_cols(w, real):
    return _SpreadCols.apply(w, real)


class _Merge357(Function):
    @staticmethod
    def forward(ctx, w3, w5, w7, b3, b5, b7, pack_dtype):
        Co, Ci = w7.shape[0], w7.shape[1]
        w = torch.empty((Co, Ci, 7, 7), dtype=torch.float32, device=w7.device)
        b = torch.empty((Co,), dtype=torch.float32, device=w7.device)
        args = (ptr(w3.detach().contiguous()), ptr(w5.detach().contiguous()), ptr(w7.detach().contiguous()), ptr(b3.detach()),
                ptr(b5.detach()), ptr(b7.detach()), ptr(w), ptr(b))
        if pack_dtype is None:
            lib().call("egm_merge357_fwd", *args, Co, Ci, stream())
            _Merge357.packs = None
        else:
            wf = torch.empty((49, pad8(Co), pad8(Ci)), dtype=pack_dtype, device=w7.device)
            wd = torch.empty((49, pad8(Ci), pad8(Co)), dtype=pack_dtype, device=w7.device)
            lib().call("egm_merge357_pack", dtype_code(pack_dtype), *args, ptr(wf), ptr(wd), Co, Ci, stream())
            _Merge357.packs = (wf, wd)
        ctx.shape = (Co, Ci)
        return w, b

    @staticmethod
    def backward(ctx, gw, gb):
        Co, Ci = ctx.shape
        dev = gw.device
        d3 = torch.empty((Co, Ci, 3, 3), dtype=torch.float32, device=dev)
        d5 = torch.empty((Co, Ci, 5, 5), dtype=torch.float32, device=dev)
        d7 = torch.empty((Co, Ci, 7, 7), dtype=torch.float32, device=dev)
        # the bias gradient goes to three parameters: handed out as three tensors (the same tensor three times makes autograd clone it
        # twice: two copy launches per FusionConv)
        gb3 = None
        if gb is not None:
            gb3 = torch.empty((3, Co), dtype=torch.float32, device=dev)
        lib().call("egm_merge357_bwd", ptr(gw.contiguous()), ptr(d3), ptr(d5), ptr(d7), ptr(gb), ptr(gb3), Co, Ci, stream())
        if gb3 is None:
            return d3, d5, d7, None, None, None, None
        return d3, d5, d7, gb3[0], gb3[1], gb3[2], None


def merge357(w3, w5, w7, b3, b5, b7, pack_dtype=None):
    w, b = _Merge357.apply(w3, w5, w7, b3, b5, b7, pack_dtype)
    if pack_dtype is not None and _Merge357.packs is not None:
        _register_pack(w, pack_dtype, *_Merge357.packs)
        _Merge357.packs = None
    return w, b


class _DwConv3(Function):
    """(depthwise3x3(x, w) + b) * scale   (RecursiveGatedAttention.dwconv and the learnable scale)"""

    @staticmethod
    def forward(ctx, x, w, b, scale):
        x, ldx = _nhwc(x)
        N, H, W, C = x.shape
        y = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
        wd, bd, sd = w.detach().contiguous(), b.detach().contiguous(), scale.detach().reshape(1).contiguous()
        lib().call("egm_dwconv3_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(wd), ptr(bd), ptr(sd), ptr(y), C, N, H, W, C, stream())
        ctx.save_for_backward(x, wd, bd, sd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, b, sc = ctx.saved_tensors
        x, ldx = _nhwc(x); g, ldg = _nhwc(g)
        N, H, W, C = x.shape
        L, dev = lib(), x.device
        dx = torch.empty(x.shape, dtype=x.dtype, device=dev)
        dw = torch.empty((C, 1, 3, 3), dtype=torch.float32, device=dev)
        db = torch.empty((C,), dtype=torch.float32, device=dev)
        ds = torch.empty((1,), dtype=torch.float32, device=dev)
        ws = torch.empty(L.query("egm_dwconv3_bwd_workspace", N, H, W, C) // 4 + 4, dtype=torch.float32, device=dev)
        L.call("egm_dwconv3_bwd", dtype_code(x.dtype), ptr(x), ldx, ptr(g), ldg, ptr(w), ptr(b), ptr(sc), ptr(dx), C, ptr(dw), ptr(db), ptr(ds),
               ptr(ws), N, H, W, C, stream())
        return dx, dw, db, ds.reshape(())


def dwconv3(x, w, b, scale):
    return _DwConv3.apply(x, w, b, scale)


# ----------------------------------------------------------------------------------------------------------
# ConvTranspose2d(k=2, s=2) of Up(bilinear=False)
# ----------------------------------------------------------------------------------------------------------
class _ConvTPack(Function):
    """[Cin][Cout][2][2] parameter -> OIHW [4*CoutP, Cin, 1, 1] matrix of the equivalent 1x1 conv (rows ordered (i, j, co))"""

    @staticmethod
    def forward(ctx, w):
        Cin, Cout = w.shape[0], w.shape[1]
        CoutP = pad8(Cout)
        w4 = torch.empty((4 * CoutP, Cin, 1, 1), dtype=torch.float32, device=w.device)
        lib().call("egm_convT2x2_pack", ptr(w.detach().contiguous()), ptr(w4), Cin, Cout, CoutP, 1, stream())
        ctx.shape = tuple(w.shape)
        return w4

    @staticmethod
    def backward(ctx, g):
        Cin, Cout = ctx.shape[0], ctx.shape[1]
        dw = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        lib().call("egm_convT2x2_pack", ptr(dw), ptr(g.contiguous()), Cin, Cout, pad8(Cout), 0, stream())
        return dw


class _Shuffle2x2(Function):
    @staticmethod
    def forward(ctx, y4, bias, C, Ho, Wo):
        y4, ld4 = _nhwc(y4)
        N, H, W, _ = y4.shape
        CP = pad8(C)
        oy, ox = (Ho - 2 * H) // 2, (Wo - 2 * W) // 2          # F.pad(x1, [dx//2, dx - dx//2, dy//2, dy - dy//2])
        out = torch.empty((N, Ho, Wo, CP), dtype=y4.dtype, device=y4.device)
        lib().call("egm_shuffle2x2_fwd", dtype_code(y4.dtype), ptr(y4), ld4, ptr(bias.detach() if bias is not None else None), C, ptr(out), CP,
                   N, H, W, CP, Ho, Wo, oy, ox, stream())
        ctx.meta = (N, H, W, CP, Ho, Wo, oy, ox, C, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, CP, Ho, Wo, oy, ox, C, has_bias = ctx.meta
        g, ldg = _nhwc(g)
        d4 = torch.empty((N, H, W, 4 * CP), dtype=g.dtype, device=g.device)
        lib().call("egm_shuffle2x2_bwd", dtype_code(g.dtype), ptr(g), ldg, ptr(d4), 4 * CP, N, H, W, CP, Ho, Wo, oy, ox, stream())
        db = None
        if has_bias and ctx.needs_input_grad[1]:
            # frame positions outside the shuffled image get no bias: sum the gradient over the image region only = over d4
            db4 = _channel_sum(d4)[0]
            db = _fold4(db4, CP)[:C]
        return d4, db, None, None, None


def _fold4(v, CP):
    """v [4*CP] fp32 -> [CP]: sum of the four (i, j) slices, on the device with the axpby kernel (no torch arithmetic)."""
    t = v.reshape(1, 1, 4, CP)
    a = _axpby(t[:, :, 0:1], 1.0, t[:, :, 1:2], 1.0)
    b = _axpby(t[:, :, 2:3], 1.0, t[:, :, 3:4], 1.0)
    return _axpby(a, 1.0, b, 1.0).reshape(CP)


def conv_transpose2x2(x, convT, out_hw):
    """nn.ConvTranspose2d(Cin, Cout, 2, stride=2) + F.pad to out_hw (src/unet.py:36,40-47).  x NHWC -> [N, Ho, Wo, pad8(Cout)]"""
    w4 = _ConvTPack.apply(convT.weight)
    y4 = conv2d(x, w4)
    return _Shuffle2x2.apply(y4, convT.bias, convT.weight.shape[1], out_hw[0], out_hw[1])


# ----------------------------------------------------------------------------------------------------------
# HEGDC pieces (src/EGM-UNet.py:210-340)
# ----------------------------------------------------------------------------------------------------------
@torch.no_grad()
def hegdc_edge_features(x, c_real):
    """The block's no_grad branch -> [N, H, W, 8] (5 real channels: scharr_x, scharr_y, sobel_x, sobel_y, blended magnitude)."""
    x, ldx = _nhwc(x)
    N, H, W, _ = x.shape
    L = lib()
    ws = torch.empty(L.query("egm_hegdc_edge_workspace", N, H, W) // 4 + 4, dtype=torch.float32, device=x.device)
    feats = torch.empty((N, H, W, 8), dtype=x.dtype, device=x.device)
    L.call("egm_hegdc_edge_features", dtype_code(x.dtype), ptr(x), ldx, c_real, ptr(feats), ptr(ws), N, H, W, stream())
    return feats


class _ScaleSigmoid(Function):
    """w * sigmoid(den): conv1_weight * Phi with Phi = phi_base (ones) * sigmoid(den)  (src/EGM-UNet.py:262-265,325-327)"""

    @staticmethod
    def forward(ctx, w, den):
        wc = w.detach().contiguous()
        out = torch.empty_like(wc)
        lib().call("egm_scale_sigmoid_fwd", ptr(wc), ptr(den.detach()), ptr(out), wc.numel(), stream())
        ctx.save_for_backward(wc, den)
        return out

    @staticmethod
    def backward(ctx, g):
        w, den = ctx.saved_tensors
        g = g.contiguous()
        dw, dden = torch.empty_like(w), torch.empty_like(den, dtype=torch.float32)
        part = _f32(256, w.device)
        lib().call("egm_scale_sigmoid_bwd", ptr(g), ptr(w), ptr(den.detach()), ptr(dw), ptr(dden), ptr(part), w.numel(), stream())
        return dw, dden.reshape(den.shape)


def scale_sigmoid(w, den):
    return _ScaleSigmoid.apply(w, den)


class _Mul2Scalar(Function):
    """a * b * alpha (alpha: learnable scalar parameter)  (src/EGM-UNet.py:332)"""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, lda = _nhwc(a); b, ldb = _nhwc(b)
        out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
        al = alpha.detach().reshape(1).float()
        lib().call("egm_mul2_scalar_fwd", dtype_code(a.dtype), ptr(a), lda, ptr(b), ldb, ptr(al), ptr(out), a.shape[3], _npix(a), a.shape[3], stream())
        ctx.save_for_backward(a, b, al)
        ctx.ashape = alpha.shape
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, al = ctx.saved_tensors
        a, lda = _nhwc(a); b, ldb = _nhwc(b); g, ldg = _nhwc(g)
        C = a.shape[3]
        da = torch.empty(a.shape, dtype=a.dtype, device=a.device)
        db = torch.empty(b.shape, dtype=b.dtype, device=b.device) if ctx.needs_input_grad[1] else None
        dalpha, part = _f32(1, a.device), _f32(1024, a.device)
        lib().call("egm_mul2_scalar_bwd", dtype_code(a.dtype), ptr(g), ldg, ptr(a), lda, ptr(b), ldb, ptr(al), ptr(da), C, ptr(db), C, ptr(dalpha),
                   ptr(part), _npix(a), C, stream())
        return da, db, dalpha.reshape(ctx.ashape)


def mul2_scalar(a, b, alpha):
    return _Mul2Scalar.apply(a, b, alpha)


# ----------------------------------------------------------------------------------------------------------
# ELA (Efficient Local Attention, src/EGM-UNet.py:56-79)
# ----------------------------------------------------------------------------------------------------------
class _ELA(Function):
    @staticmethod
    def forward(ctx, x, conv_w, gamma, beta, groups, eps):
        x, ldx = _nhwc(x)
        N, H, W, C = x.shape
        L, dt, st, dev = lib(), dtype_code(x.dtype), stream(), x.device
        ks = conv_w.shape[-1]
        cw = conv_w.detach().reshape(C, ks).contiguous()
        mh, mw = _f32((N, H, C), dev), _f32((N, W, C), dev)
        L.call("egm_ela_strip_means", dt, ptr(x), ldx, ptr(mh), ptr(mw), N, H, W, C, st)
        yh, yw, gh, gw = _f32((N, H, C), dev), _f32((N, W, C), dev), _f32((N, H, C), dev), _f32((N, W, C), dev)
        stats = _f32((N, 2, groups, 2), dev)
        L.call("egm_ela_gates_fwd", ptr(mh), ptr(mw), ptr(cw), ks, ptr(gamma.detach()), ptr(beta.detach()), float(eps), ptr(yh), ptr(yw), ptr(gh),
               ptr(gw), ptr(stats), N, H, W, C, groups, st)
        out = torch.empty((N, H, W, C), dtype=x.dtype, device=dev)
        L.call("egm_ela_apply", dt, ptr(x), ldx, ptr(gh), ptr(gw), ptr(out), C, N, H, W, C, st)
        ctx.save_for_backward(x, mh, mw, yh, yw, gh, gw, stats, cw, gamma)
        ctx.meta = (groups, ks, conv_w.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, mh, mw, yh, yw, gh, gw, stats, cw, gamma = ctx.saved_tensors
        groups, ks, wshape = ctx.meta
        x, ldx = _nhwc(x); g, ldg = _nhwc(g)
        N, H, W, C = x.shape
        L, dt, st, dev = lib(), dtype_code(x.dtype), stream(), x.device
        ws = torch.empty(L.query("egm_ela_bwd_workspace", N, H, W, C, ks) // 4 + 4, dtype=torch.float32, device=dev)
        dx = torch.empty((N, H, W, C), dtype=x.dtype, device=dev)
        dcw, dgamma, dbeta = _f32((C, ks), dev), _f32(C, dev), _f32(C, dev)
        L.call("egm_ela_bwd", dt, ptr(g), ldg, ptr(x), ldx, ptr(mh), ptr(mw), ptr(yh), ptr(yw), ptr(gh), ptr(gw), ptr(stats), ptr(cw), ks,
               ptr(gamma.detach()), ptr(dx), C, ptr(dcw), ptr(dgamma), ptr(dbeta), ptr(ws), N, H, W, C, groups, st)
        return dx, dcw.reshape(wshape), dgamma, dbeta, None, None


def ela(x, conv, gn):
    """conv: nn.Conv1d(C, C, k, padding=k//2, groups=C, bias=False); gn: nn.GroupNorm(G, C) -- parameter holders"""
    return _ELA.apply(x, conv.weight, gn.weight, gn.bias, gn.num_groups, gn.eps)


# ----------------------------------------------------------------------------------------------------------
# MCALayer
# ----------------------------------------------------------------------------------------------------------
class _MCALayer(Function):
    """wc / kc None: MCALayer(no_spatial=True) -- two gates, x_out = x*(g_h+g_w)/2 (src/EGM-UNet.py:766-771)"""

    @staticmethod
    def forward(ctx, x, x_coef, x_act, wh, kh, ww, kw, wc, kc, training, exclusive=False):
        """x_coef / x_act: x is a Lazy's stand-in (the raw output of the conv in front; ops.Lazy): its BatchNorm + activation is applied
        by the statistics pass, which writes the materialised tensor on the way (egm_mca_reduce_bn)."""
        x, ldx = _nhwc(x)
        N, H, W, C = x.shape
        L, dt, st, dev = lib(), dtype_code(x.dtype), stream(), x.device
        Lax = H + W + C
        ws = torch.empty(L.query("egm_mca_reduce_workspace", N, H, W, C) // 4 + 4, dtype=torch.float32, device=dev)
        sums = _f32((N, Lax, 2), dev)
        if x_coef is not None:
            z = torch.empty((N, H, W, C), dtype=x.dtype, device=dev)
            L.call("egm_mca_reduce_bn", dt, ptr(x), ldx, ptr(x_coef[0]), ptr(x_coef[1]), x_act, ptr(z), C, ptr(sums), ptr(ws), N, H, W, C, st)
            x, ldx = z, C
        else:
            L.call("egm_mca_reduce", dt, 0, ptr(x), ldx, None, 0, ptr(sums), ptr(ws), N, H, W, C, st)
        stats, o, gates = _f32((N, Lax, 2), dev), _f32((N, Lax), dev), _f32((N, Lax), dev)
        ps = [t.detach().contiguous() for t in (wh, kh, ww, kw, wc, kc) if t is not None]
        ns = int(wc is None)
        ks = (kh.numel(), kw.numel(), 0 if ns else kc.numel())
        pc = (None, None) if ns else (ptr(ps[4]), ptr(ps[5]))
        L.call("egm_mca_gates_fwd", ptr(sums), ptr(ps[0]), ptr(ps[1]), ks[0], ptr(ps[2]), ptr(ps[3]), ks[1], pc[0], pc[1], ks[2],
               ptr(stats), ptr(o), ptr(gates), N, H, W, C, st)
        # x_out = x * gate, the 3x3 stencils (range, squared high-pass, its average), the channel shuffle and the sum: one fused pass
        xo = torch.empty((N, H, W, C), dtype=x.dtype, device=dev) if training else None
        codes = torch.empty((N, H, W, C), dtype=torch.uint8, device=dev) if training else None
        out = torch.empty((N, H, W, C), dtype=x.dtype, device=dev)
        L.call("egm_mca_fused_fwd", dt, ptr(x), ldx, ptr(gates), ptr(xo), C, ptr(out), C, ptr(codes), N, H, W, C, ns, st)
        if training:
            ctx.save_for_backward(x, xo, codes, stats, o, gates, *ps)
            ctx.ks = ks
            # exclusive: the caller vouches that the conv -> BatchNorm stand-in this layer materialised feeds nothing else, so the last
            # backward step (dx) can be left to that BatchNorm's backward (see _defer_dz)
            ctx.defer_dx = bool(exclusive and x_coef is not None and _dz_fusable(C))
        return out

    @staticmethod
    def backward(ctx, g):
        x, xo, codes, stats, o, gates, wh, kh, ww, kw, *pc = ctx.saved_tensors
        ks = ctx.ks
        ns = int(not pc)
        x, ldx = _nhwc(x); g, ldg = _nhwc(g)
        N, H, W, C = x.shape
        L, dt, st, dev = lib(), dtype_code(x.dtype), stream(), x.device
        Lax = H + W + C
        dxo = torch.empty_like(xo)
        if _FUSE_MCA_BWD and x.dtype == torch.bfloat16:
            # du and dxo in one tiled pass: the 3 x 3 neighbourhoods come from LDS, du never reaches memory
            L.call("egm_mca_bwd_dudxo", dt, ptr(codes), ptr(xo), C, ptr(g), ldg, ptr(dxo), C, N, H, W, C, st)
        else:
            du = torch.empty_like(xo)
            L.call("egm_mca_bwd_du", dt, ptr(xo), C, ptr(g), ldg, ptr(du), C, N, H, W, C, st)
            L.call("egm_mca_bwd_dxo", dt, ptr(codes), ptr(g), ldg, ptr(du), C, ptr(dxo), C, N, H, W, C, st)
        ws = torch.empty(L.query("egm_mca_reduce_workspace", N, H, W, C) // 4 + 4, dtype=torch.float32, device=dev)
        dG = _f32((N, Lax, 2), dev)
        L.call("egm_mca_reduce", dt, 1, ptr(dxo), C, ptr(x), ldx, ptr(dG), ptr(ws), N, H, W, C, st)
        dz = _f32((N, Lax), dev)
        coef = _f32((N, Lax, 2), dev)
        dwts = _f32((3, 2), dev)
        dks = _f32((3, 8), dev)                                  # zeroed by the kernel itself
        L.call("egm_mca_gates_bwd", ptr(dG), ptr(stats), ptr(o), ptr(gates), ptr(wh), ptr(kh), ks[0], ptr(ww), ptr(kw), ks[1],
               None if ns else ptr(pc[0]), None if ns else ptr(pc[1]), ks[2], ptr(dz), ptr(coef), ptr(dwts), ptr(dks), N, H, W, C, st)
        if ctx.defer_dx and _FUSE_DZ:
            dx = dxo                                            # stand-in: the BatchNorm backward in front evaluates dx per vector itself
            _defer_dz(dxo, ("mca", dxo, C, gates, coef, ns, (N, H, W)))
        else:
            dx = torch.empty((N, H, W, C), dtype=x.dtype, device=dev)
            L.call("egm_mca_bwd_dx", dt, ptr(dxo), C, ptr(x), ldx, ptr(gates), ptr(coef), ptr(dx), C, N, H, W, C, ns, st)
        gk = [dks[a, :ks[a]].reshape(1, 1, 1, ks[a]) for a in range(2 if ns else 3)]
        if ns:
            return dx, None, None, dwts[0], gk[0], dwts[1], gk[1], None, None, None, None
        return dx, None, None, dwts[0], gk[0], dwts[1], gk[1], dwts[2], gk[2], None, None


def mca_layer(x, layer, training, exclusive=False):
    """layer: an MCALayer parameter holder with gates h_cw, w_hc and -- unless layer.no_spatial -- c_hw
    (each: .weight [2], .conv.weight [1,1,1,k])."""
    c = (None, None) if layer.no_spatial else (layer.c_hw.weight, layer.c_hw.conv.weight)
    xt, xc, xa = _unlazy(x)                   # a Lazy (conv_bn_lazy) is materialised by the layer's own statistics pass
    C = xt.shape[3]
    if xc is not None and not (8 <= C <= 512 and (C & (C - 1)) == 0):
        xt, xc, xa = materialize(x), None, ACT_NONE
    return _MCALayer.apply(xt, xc, xa, layer.h_cw.weight, layer.h_cw.conv.weight, layer.w_hc.weight, layer.w_hc.conv.weight, c[0], c[1],
                           training, exclusive and xc is not None)


class _SAConv7(Function):
    """SpatialAttentionModule.conv1 (7x7, 2->1) on the (mean, max) map as a direct stencil."""

    @staticmethod
    def forward(ctx, x, w):
        x, ldx = _nhwc(x)
        N, H, W, _ = x.shape
        wd = w.detach().contiguous()
        y = torch.empty((N, H, W, 8), dtype=x.dtype, device=x.device)
        lib().call("egm_sa_conv7_fwd", dtype_code(x.dtype), ptr(x), ldx, ptr(wd), ptr(y), 8, N, H, W, stream())
        ctx.save_for_backward(x, wd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        x, ldx = _nhwc(x); g, ldg = _nhwc(g)
        N, H, W, _ = x.shape
        L, dev = lib(), x.device
        dx = torch.empty((N, H, W, 8), dtype=x.dtype, device=dev)
        dw = torch.empty((1, 2, 7, 7), dtype=torch.float32, device=dev)
        ws = torch.empty(L.query("egm_sa_conv7_bwd_workspace", N, H, W) // 4 + 4, dtype=torch.float32, device=dev)
        L.call("egm_sa_conv7_bwd", dtype_code(x.dtype), ptr(x), ldx, ptr(g), ldg, ptr(w), ptr(dx), 8, ptr(dw), ptr(ws), N, H, W, stream())
        return dx, dw


def sa_conv7(x, w):
    return _SAConv7.apply(x, w)
