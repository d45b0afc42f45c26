#!/usr/bin/env python3
"""Headline benchmark: EGM-UNet training throughput (BASELINE.json: "train images/sec at 3x512x512 bs=8/GPU").

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or plainly as
    `python bench.py --gpus N`, which then starts that launcher itself as a child process)

One step = forward + 5-term criterion + backward + fused SGD of GRFBUNet(3, 2, base_c=32) on a device-resident synthetic
batch of 8 x 3 x 512 x 512 per GPU, bf16 activation storage / MFMA, fp32 accumulation and master weights.
Prints ONE JSON line (rank 0) with the extra `roofline` (dominant kernel, timed with HIP events on the launch stream in an
instrumented step outside the timed region) and `cpu_baseline` (CPU oracle on a bounded sample) objects.
"""
import argparse
import struct
import contextlib
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def synth_batch(n, h, w, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, h, w, generator=g)
    t = torch.zeros(n, h, w, dtype=torch.int64)
    for i in range(n):                                   # one random quadrilateral-ish blob of foreground (~15 %)
        cy, cx = torch.randint(h // 4, 3 * h // 4, (2,), generator=g).tolist()
        hh, ww = torch.randint(h // 6, h // 3, (2,), generator=g).tolist()
        t[i, max(0, cy - hh // 2):cy + hh // 2, max(0, cx - ww // 2):cx + ww // 2] = 1
    t[torch.rand(n, h, w, generator=g) < 0.01] = 255     # 1 % ignore pixels
    return x.to(device), t.to(device)


class _Span:
    """elapsed_time() of an event pair divided by the number of identical launches it brackets"""

    def __init__(self, e0, e1, reps):
        self.e0, self.e1, self.reps = e0, e1, reps

    def elapsed_time(self, _unused=None):
        return self.e0.elapsed_time(self.e1) / self.reps


class KernelTimer:
    """Times every C-ABI call with HIP events recorded on the stream the kernels are launched on (torch's current
    stream, which is the stream handed to the library)."""

    def __init__(self, lib):
        self.lib, self.records, self._orig, self.grouped = lib, [], lib.call, []

    REPS = 4          # conv launches are re-issued back to back so the event pair brackets kernel time, not launch gaps

    def __enter__(self):
        def timed(name, *args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 1
            if name in ("egm_conv_fwd", "egm_conv_wgrad", "egm_conv_wgrad_multi"):
                # idempotent (same inputs, outputs overwritten with the same values): the first call does the work of the step,
                # REPS more are timed as a train, which matches the per-launch durations rocprofv3 reports for the graph replay
                self._orig(name, *args)
                reps = self.REPS
            e0.record()
            for _ in range(reps):
                self._orig(name, *args)
            e1.record()
            from egm_unet_amd import ops as _o
            self.records.append((name, args, _Span(e0, e1, reps), None))
            self.grouped.append(_o.conv_group.serial if _o.conv_group.depth > 0 else 0)      # which launch group the call sits in (0 = none)
        self.lib.call = timed
        return self

    def __exit__(self, *a):
        self.lib.call = self._orig
        torch.cuda.synchronize()

    def conv_table(self):
        """Per conv shape: launches, time, achieved TFLOP/s and algorithmic GB/s (bf16 in+out+weights once)."""
        rows = {}
        for name, args, e0, e1 in self.records:
            if name == "egm_conv_fwd":
                N, H, W, Cin, Cout, KH, KW, dil = args[9:17]
                kind = "fwd/dgrad"
            elif name == "egm_conv_wgrad":
                N, H, W, Cin, Cout = args[7:12]
                KH, KW, dil = args[14:17]
                kind = "wgrad"
            elif name == "egm_conv_wgrad_multi":
                # one launch of several layers: its time is shared out by the members' roofline times
                mem = self._wgrad_members(args)
                roofs = [max(2.0 * m[0] * m[1] * m[2] * m[3] * m[4] * m[5] * m[6] / (MFMA_BF16_PEAK_TFLOPS * 1e12),
                             2.0 * m[0] * m[1] * m[2] * (m[3] + m[4]) / (HBM_PEAK_GBS * 1e9)) for m in mem]
                for m, rf in zip(mem, roofs):
                    r = rows.setdefault(("wgrad*", m[5], m[7], m[3], m[4], m[1], m[2]), [0, 0.0])
                    r[0] += 1; r[1] += e0.elapsed_time(e1) * rf / sum(roofs)
                continue
            else:
                continue
            key = (kind, KH, dil, Cin, Cout, H, W)
            r = rows.setdefault(key, [0, 0.0])
            r[0] += 1; r[1] += e0.elapsed_time(e1)
        out = []
        for (kind, KH, dil, Cin, Cout, H, W), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            flop = 2.0 * 8 * H * W * Cin * Cout * KH * KH
            byts = 2.0 * (8 * H * W * (Cin + Cout) + KH * KH * Cin * Cout)
            t = ms / n * 1e-3
            out.append(f"{kind:9s} k{KH} d{dil:<2d} {Cin:4d}->{Cout:<4d} {H:3d}x{W:<3d} n={n:2d} avg={ms / n * 1e3:7.1f}us "
                       f"{flop / t / 1e12:7.1f} TF/s {byts / t / 1e9:7.0f} GB/s  total={ms:6.3f}ms")
        return out

    ENCODER_3X3 = [("in_conv.3", 32, 32, 512), ("down1.1.0", 32, 64, 256), ("down1.1.4", 64, 64, 256), ("down2.1.0", 64, 128, 128),
                   ("down2.1.4", 128, 128, 128), ("down3.1.0", 128, 256, 64), ("down3.1.4", 256, 256, 64), ("down4.1.0/4", 256, 256, 32)]

    def encoder_table(self):
        """The 3x3 encoder convs north_star names, by shape (forward and data-gradient launches of that shape pooled): achieved
        TFLOP/s, algorithmic GB/s and the fraction of the layer's own roofline min(MFMA peak, arithmetic intensity x HBM peak)."""
        out = []
        for name, cin, cout, hw in self.ENCODER_3X3:
            n, ms, N = 0, 0.0, 8
            for rname, args, e0, e1 in self.records:
                if rname != "egm_conv_fwd":
                    continue
                N_, H, W, Cin, Cout, KH, KW, dil = args[9:17]
                if KH == 3 and dil == 1 and H == hw and ((Cin, Cout) == (cin, cout) or (Cin, Cout) == (cout, cin)):
                    n += 1; ms += e0.elapsed_time(e1); N = N_
            if not n:
                continue
            flop = 2.0 * N * hw * hw * cin * cout * 9
            byts = 2.0 * (N * hw * hw * (cin + cout) + 9 * cin * cout)
            t = ms / n * 1e-3
            roof_t = max(flop / (MFMA_BF16_PEAK_TFLOPS * 1e12), byts / (HBM_PEAK_GBS * 1e9))
            out.append({"layer": name, "shape": f"{cin}->{cout}@{hw}", "launches": n, "us": round(t * 1e6, 1), "tflops": round(flop / t / 1e12, 1),
                        "gbs": round(byts / t / 1e9), "bound": "mfma" if flop / (MFMA_BF16_PEAK_TFLOPS * 1e12) > byts / (HBM_PEAK_GBS * 1e9) else "hbm",
                        "frac_of_layer_roofline": round(roof_t / t, 3)})
        return out

    _WGRAD_DESC = struct.Struct("<3Q14i")        # egm_conv_wgrad_desc (include/egm_hip.h)

    def _wgrad_members(self, args):
        """egm_conv_wgrad_multi(dtype, descs, n, stream) -> [(N, H, W, Cin, Cout, KH, KW, dil)] of its members"""
        blob, n = args[1], args[2]
        out = []
        for i in range(n):
            f = self._WGRAD_DESC.unpack_from(blob, i * self._WGRAD_DESC.size)
            out.append((f[5], f[6], f[7], f[8], f[9], f[12], f[13], f[14]))
        return out

    def _multi_name(self, args):
        """kernel of an egm_conv_wgrad_multi call as a trace prints it: the members' kernel, its merged form when there are several"""
        mem = self._wgrad_members(args)
        k = self._kernel_name("egm_conv_wgrad_kernel_name", args[0], *mem[0][:5], *mem[0][5:8])
        return k.replace("_kernel<", "_multi_kernel<") if len(mem) > 1 else k

    def _kernel_name(self, entry, dtype, N, H, W, Cin, Cout, KH, KW, dil):
        """The kernel a conv call takes, spelled as in a rocprofv3 kernel trace (egm_conv_kernel_name / egm_conv_wgrad_kernel_name), so
        that the per-launch averages below sit beside the matching rows of profiles/*_kernel_trace_summary.md."""
        buf = ctypes.create_string_buffer(96)
        getattr(self.lib.cdll, entry)(dtype, N, H, W, Cin, Cout, KH, KW, dil, ctypes.cast(buf, ctypes.c_void_p), 96)
        return buf.value.decode()

    FAMILIES = (("conv fwd/dgrad", ("egm_conv_fwd", "egm_conv_fwd_split", "egm_group_end")),
                ("weight gradients", ("egm_conv_wgrad", "egm_wgrad_reduce", "egm_sa_conv7_bwd_w", "egm_dwconv3_bwd_w")),
                ("1x1 backward, dx + dW in one pass", ("egm_conv1x1_bwd",)),
                ("pointwise conv+BatchNorm (moment form)", ("egm_pw_",)),
                ("BatchNorm", ("egm_bn_", "egm_channel_sums", "egm_reduce_tiles")),
                ("MCALayer", ("egm_mca_",)))

    def families(self):
        """ms and C-ABI calls per kernel family of the instrumented step (HIP-event pair per call; conv calls as trains of REPS)."""
        fam = {}
        for name, _args, e0, e1 in self.records:
            key = next((f for f, pre in self.FAMILIES if name.startswith(pre)), "other")
            a = fam.setdefault(key, [0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1)
        return {k: {"calls": v[0], "ms": round(v[1], 3)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}

    def summary(self):
        """-> {kernel: [launches, ms, flop, bytes, roofline ms, ungrouped launches, ungrouped bytes]}.  'Ungrouped' = issued outside an
        ops.conv_group block: only those run under THIS kernel name in the replayed graph (grouped members run as one *_multi launch),
        so they are the launches a rocprofv3 trace / PMC pass of the graph replays attributes to it."""
        agg = {}
        keyed = []
        for name, args, e0, e1 in self.records:
            k = name
            if name == "egm_conv_fwd":
                k = self._kernel_name("egm_conv_kernel_name", args[0], *args[9:17])
            elif name == "egm_conv_wgrad":
                k = self._kernel_name("egm_conv_wgrad_kernel_name", args[0], *args[7:12], *args[14:17])
            elif name == "egm_conv_wgrad_multi":
                k = self._multi_name(args)
            keyed.append(k)
        # a launch group merges the members that run the SAME kernel instantiation; a member alone of its kind launches under its own name
        members = {}
        for k, gid in zip(keyed, self.grouped):
            if gid:
                members[(gid, k)] = members.get((gid, k), 0) + 1
        for (name, args, e0, e1), gid, kk in zip(self.records, self.grouped, keyed):
            grouped = bool(gid) and members[(gid, kk)] > 1
            key = name
            flops = 0.0
            if name == "egm_conv_fwd":
                # (dtype, x, ldx, wf, bias, bias_n, y, ldy, stats, N, H, W, Cin, Cout, KH, KW, dil, stream)
                N, H, W, Cin, Cout, KH, KW, dil = args[9:17]
                flops = 2.0 * N * H * W * Cin * Cout * KH * KW
                key = self._kernel_name("egm_conv_kernel_name", args[0], N, H, W, Cin, Cout, KH, KW, dil)
            elif name == "egm_conv_wgrad":
                N, H, W, Cin, Cout = args[7:12]
                KH, KW, dil = args[14:17]
                flops = 2.0 * N * H * W * Cin * Cout * KH * KW
                key = self._kernel_name("egm_conv_wgrad_kernel_name", args[0], N, H, W, Cin, Cout, KH, KW, dil)
            elif name == "egm_conv_wgrad_multi":
                # ONE launch (ops._launch_pending_slabs hands over members of one kernel instantiation): flop, bytes and the roofline
                # time of the launch are the sums over its members
                key = self._multi_name(args)
                a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0, 0, 0.0])
                a[0] += 1; a[1] += e0.elapsed_time(e1); a[5] += 1
                for N, H, W, Cin, Cout, KH, KW, dil in self._wgrad_members(args):
                    fl = 2.0 * N * H * W * Cin * Cout * KH * KW
                    byts = 2.0 * (N * H * W * (Cin + Cout) + KH * KW * Cin * Cout)          # as for a single launch: operands once each
                    a[2] += fl; a[3] += byts; a[6] += byts
                    a[4] += 1e3 * max(fl / (MFMA_BF16_PEAK_TFLOPS * 1e12), byts / (HBM_PEAK_GBS * 1e9))
                continue
            a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0, 0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flops
            if flops:
                byts = 2.0 * (N * H * W * (Cin + Cout) + KH * KW * Cin * Cout)        # bf16 in + out + weights, once each
                a[3] += byts
                # per-launch roofline time: the layer is bound by whichever of MFMA peak and HBM peak takes longer
                a[4] += 1e3 * max(flops / (MFMA_BF16_PEAK_TFLOPS * 1e12), byts / (HBM_PEAK_GBS * 1e9))
                if not grouped:
                    a[5] += 1; a[6] += byts
        return agg


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle (oracle/*.py, PyTorch fp32, the reference's algorithm) on a bounded sample of the same workload:
    train steps (fwd + criterion + bwd + SGD) of EGM-UNet(3,2,32) at bs 2 x 3 x 512 x 512."""
    from oracle import egm_ref as R, loss_ref as L
    threads = torch.get_num_threads()
    st = R.make_egm_unet_state(3, 2, 32, seed=0)
    params = {k: v.clone() for k, v in st.items() if v.is_floating_point() and "running_" not in k}
    bufs, lw = {}, torch.tensor([1.0, 2.0])
    x, t = synth_batch(2, 512, 512, 0, "cpu")
    times = []
    t_start = time.time()
    for step in range(3):
        t0 = time.time()
        work = dict(st)
        for k in params:
            work[k] = params[k].detach().clone().requires_grad_(True)
        loss = L.criterion(R.egm_unet_forward(work, x, True), t, lw, num_classes=2, ignore_index=255)
        loss.backward()
        with torch.no_grad():
            L.sgd_step(params, {k: work[k].grad for k in params}, bufs, lr=0.02)
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget:
            break
    best = min(times[1:]) if len(times) > 1 else times[0]
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(2.0 / best, 4), "unit": "images/s", "cores": threads, "cpu_model": cpu_model, "kind": "port",
            "sample": f"{len(times)} train steps (fwd+loss+bwd+SGD) of EGM-UNet(3,2,32) at bs 2x3x512x512 fp32 on the CPU oracle; best of steps after the first"}


def _rccl_version():
    """RCCL's version as torch reports it (torch.cuda.nccl IS RCCL on ROCm); NCCL_DEBUG=VERSION makes RCCL print the same at init."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(i) for i in v) if isinstance(v, tuple) else str(v)
    except Exception as e:                                      # noqa: BLE001 -- a report field must not cost the measurement
        return f"unavailable ({type(e).__name__})"


def _seeded_clipseg(dev, dtype, seed=0):
    """CLIPDensePredT('ViT-B/16', reduce_dim=64) with seeded random weights of the reference's architecture (it ships none)."""
    from egm_unet_amd.clipseg import CLIPDensePredT
    m = CLIPDensePredT(version="ViT-B/16", reduce_dim=64)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith(("ln_1.weight", "ln_2.weight", "ln_pre.weight", "ln_post.weight", "ln_final.weight", "norm1.weight", "norm2.weight")):
                p.fill_(1.0)
            elif p.dim() >= 2 or "embedding" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            else:
                p.zero_()
    return m.to(dev).set_compute_dtype(torch.bfloat16 if dtype == "bf16" else torch.float32)


def clipseg_bench(args, dev, rank, world):
    """BASELINE.json configs[3]: CLIPDensePredT('ViT-B/16', reduce_dim=64) on 352x352, seeded synthetic weights, bf16.
    clipseg_infer: forward with prompts encoded per call (uncached).  clipseg_train: frozen-backbone forward + decoder forward,
    BCE-with-logits, decoder backward, AdamW (experiments/phrasecut.yaml, batch 64).  Replicas only (no exchange)."""
    train = args.workload == "clipseg_train"
    m = _seeded_clipseg(dev, args.dtype)
    B = args.batch if args.batch != 8 else (64 if train else 32)      # experiments/phrasecut.yaml batch sizes
    x = torch.randn(B, 3, 352, 352, generator=torch.Generator().manual_seed(rank)).to(dev)
    prompts = ["a photo of a tactile paving."] * B
    if train:
        from egm_unet_amd.clip import train_ops as T
        m.train()
        opt = T.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        target = (torch.rand(B, 1, 352, 352, generator=torch.Generator().manual_seed(100 + rank)) < 0.3).float().to(dev)
        cond = m.compute_conditional(prompts)                         # prompts are constant across iterations: encoded once

        def step():
            loss = T.bce_with_logits(m(x, cond)[0], target)
            opt.zero_grad(); loss.backward(); opt.step()
    else:
        m.eval()

        def step():
            m(x, prompts)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if rank == 0:
        what = ("decoder training step (frozen ViT-B/16 forward + decoder fwd/bwd + BCE + AdamW)" if train
                else "forward (text encoder uncached)")
        print(json.dumps({"metric": f"CLIPSeg {'decoder training' if train else 'inference'} images/sec at 3x352x352",
                          "value": round(B * world * args.steps / el, 2), "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": f"CLIPDensePredT ViT-B/16 rd64 {what}, {B}x3x352x352 + {B} prompts",
                                     "global_batch": B * world, "parallelism": f"replicas x{world}"}}))


def ablation_bench(args, dev, rank, world):
    """BASELINE.json configs[4]: edge-guided attention + GRFB ablation at 3x1024x1024.
    (a) the HBM-bound blocks in isolation on shapes derived from a Bx3x1024x1024 input with base_c = 32 (forward and backward timed
    separately with HIP events on the launch stream); their GB/s is ALGORITHMIC bytes (inputs + outputs once each, bf16; backward:
    incoming gradient + saved input + outgoing gradient) over the measured time, i.e. the fraction of a perfectly fused
    implementation's HBM roofline.  (b) the full EGM-UNet(3,2,32) train step on Bx3x1024x1024 through the hipGraph path."""
    from egm_unet_amd import GRFBUNet, ops
    from egm_unet_amd._lib import require_gpu
    from egm_unet_amd.egm_unet import Down, EdgeAwareFeatureEnhancer, EdgeEnhancedGRFB, MCALayer
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.train_utils import criterion
    from egm_unet_amd.unet import Up
    require_gpu()
    B = args.batch if args.batch != 8 else 2
    S = args.size if args.size != 512 else 1024
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    esz = 2 if args.dtype == "bf16" else 4
    g = torch.Generator().manual_seed(7 + rank)
    reps = 10

    def nhwc(c, s):
        return torch.randn(B, s, s, c, generator=g).to(dev).to(dt)

    # everything of part (a) lives on ONE non-default stream: tensors, autograd accumulators, warm-up, capture and replay
    # (an AccumulateGrad node tied to another stream would put a cross-stream wait into the capture)
    work = torch.cuda.Stream()

    def timed(fn):
        """mean device time of fn() replayed from a captured hipGraph (eager launches of the multi-kernel blocks are host-bound)"""
        for _ in range(2):
            fn()
        work.synchronize()
        graph = torch.cuda.CUDAGraph()
        import torch.distributed as _d
        with torch.cuda.graph(graph, stream=work, capture_error_mode="thread_local" if _d.is_initialized() else "global"):   # RCCL watchdog thread, see graph.py
            fn()
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(work)
        for _ in range(reps):
            graph.replay()
        e1.record(work)
        work.synchronize()
        return e0.elapsed_time(e1) / reps

    def block(name, row, module, ins, bytes_fwd=None, bytes_bwd=None):
        """forward / backward of one block; algorithmic bytes default to in + out (fwd) and gout + in + gin (bwd)"""
        if isinstance(module, torch.nn.Module):
            module.to(dev).train()
        xs = [x.clone().requires_grad_(True) for x in ins]
        with torch.no_grad():
            out = module(*ins)
        go = torch.ones_like(out)
        nin, nout = sum(x.numel() for x in ins), out.numel()
        bf = bytes_fwd if bytes_fwd is not None else (nin + nout) * esz
        bb = bytes_bwd if bytes_bwd is not None else (nout + 2 * nin) * esz
        def f():
            with torch.no_grad():
                module(*ins)
        t_f = timed(f)

        def fb():
            for x in xs:
                x.grad = None
            module(*xs).backward(go)
        t_fb = timed(fb)
        t_b = max(t_fb - t_f, 1e-6)
        return {"block": name, "row": row, "fwd_ms": round(t_f, 4), "bwd_ms": round(t_b, 4),
                "fwd_GBs": round(bf / t_f / 1e6, 1), "bwd_GBs": round(bb / t_b / 1e6, 1),
                "fwd_frac_hbm": round(bf / t_f / 1e6 / HBM_PEAK_GBS, 3), "bwd_frac_hbm": round(bb / t_b / 1e6 / HBM_PEAK_GBS, 3)}

    torch.manual_seed(0)
    rows = []
    h = S // 2
    torch.cuda.synchronize()
    _ctx = torch.cuda.stream(work)
    _ctx.__enter__()
    rows.append(block("MaxPool2d(2) 32ch @%d" % S, "K3", ops.maxpool2, [nhwc(32, S)]))
    rows.append(block("bilinear x2 + concat (32ch @%d skip, 32ch @%d low)" % (S, h), "K10", lambda sk, lo: ops.upcat(sk, lo),
                      [nhwc(32, S), nhwc(32, h)]))
    # the product path: the skip tensor's producer wrote it into the concat buffer, the kernel adds only the upsampled half
    cbuf, (cslot, _) = ops.cat_slots(B, S, S, [32, 32], dt, dev)
    low_in = nhwc(32, h)

    def upcat_inplace():
        with torch.no_grad():
            ops.upcat(cslot, low_in, cbuf)
    t_ip = timed(upcat_inplace)
    b_ip = (low_in.numel() + cbuf.numel() // 2) * esz
    rows.append({"block": "bilinear x2 into the concat buffer that already holds the skip (32ch @%d low -> @%d)" % (h, S), "row": "K10 in place",
                 "fwd_ms": round(t_ip, 4), "fwd_GBs": round(b_ip / t_ip / 1e6, 1), "fwd_frac_hbm": round(b_ip / t_ip / 1e6 / HBM_PEAK_GBS, 3)})
    del cbuf, cslot, low_in
    rows.append(block("MCALayer(64) @%d" % h, "K4", MCALayer(64), [nhwc(64, h)]))
    rows.append(block("EdgeAwareFeatureEnhancer(64) @%d" % h, "K5", EdgeAwareFeatureEnhancer(64), [nhwc(64, h)]))
    rows.append(block("EdgeEnhancedGRFB(64,64) @%d" % h, "K5-K8", EdgeEnhancedGRFB(64, 64), [nhwc(64, h)]))
    rows.append(block("Down(32,64) from @%d" % S, "K3+K1/K2+K4+K5-K8", Down(32, 64), [nhwc(32, S)]))
    rows.append(block("Up(64,32) @%d" % S, "K10+K1/K2", Up(64, 32), [nhwc(32, h), nhwc(32, S)]))
    work.synchronize()
    _ctx.__exit__(None, None, None)
    if rank == 0:
        for r in rows:
            print("[ablation] " + json.dumps(r), file=sys.stderr)
    torch.cuda.empty_cache()

    # (b) the whole network at 1024 x 1024
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):           # the constructor prints a line like the reference's (src/EGM-UNet.py:516); stdout carries the JSON line only
        model = GRFBUNet(3, 2, base_c=32).to(dev).train()
    model.set_compute_dtype(dt)
    opt = SGD(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    x, t = synth_batch(B, S, S, 1000 + rank, dev)
    lw = torch.tensor([1.0, 2.0], device=dev)
    from egm_unet_amd.graph import GraphedTrainStep
    step = GraphedTrainStep(model, opt, x, t, lw, num_classes=2, ignore_index=255, reducer=None, warmup=2)
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if rank == 0:
        k10 = rows[1]
        print(json.dumps({
            "metric": f"train images/sec at 3x{S}x{S} bs={B}/GPU (ablation, BASELINE.json configs[4])", "value": round(B * world * args.steps / el, 3),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"EGM-UNet GRFBUNet(3,2,base_c=32) fwd+loss+bwd+SGD, {B}x3x{S}x{S} per GPU + block table (replicas, no exchange)",
                       "global_batch": B * world, "launch": "hipGraph replay", "final_loss": round(float(loss.detach()), 4)},
            "roofline": {"bound": "hbm", "achieved": k10["fwd_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k10["fwd_frac_hbm"],
                         "traffic": None, "kernel": "egm_upcat_fwd (K10 bilinear x2 + pad + concat, one kernel)"},
            "blocks": rows}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="egm_unet_train", choices=["egm_unet_train", "clipseg_infer", "clipseg_train", "ablation_1024"],
                    help="egm_unet_train = the headline metric (BASELINE.json configs[1]); clipseg_infer = configs[3] (ViT-B/16 image+text "
                         "encode + decoder on 352x352), reported as a secondary line")
    ap.add_argument("--instrument", action="store_true", help="N > 1: also run the instrumented eager step (roofline / families objects) on every rank")
    ap.add_argument("--eager", action="store_true", help="issue every kernel from Python instead of replaying the captured hipGraph")
    ap.add_argument("--dp-single-graph", action="store_true",
                    help="N > 1: one captured graph (forward + backward) followed by the exchange and the SGD launch, instead of the three-graph "
                         "step that overlaps bucket 0's all-reduce with the encoder backward (fallback for A/B and triage)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves as a CHILD process (one rank per GPU over
        # torch.distributed.run), before this process has touched the GPU, pass their output through (rank 0 prints the JSON
        # line on the inherited stdout) and exit with the launcher's return code.  Never exec: the parent stays a plain waiter.
        import subprocess
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("NCCL_DEBUG", "VERSION")                # RCCL prints its version once at init (stderr)
        # --standalone: the launcher binds its rendezvous store to a free port itself (picking one here and closing the socket first
        # can collide on a busy host); --local-addr because the container hostname may not resolve
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node",
               str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env, cwd=ROOT).returncode)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launched by torch.distributed.run with another --nproc-per-node?)")
    if os.environ.get("EGM_BENCH_SINGLE_DEVICE"):          # rehearsal of the N>1 code path on a one-GPU box (with EGM_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EGM_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.workload in ("clipseg_infer", "clipseg_train"):
        return clipseg_bench(args, dev, rank, world)
    if args.workload == "ablation_1024":
        return ablation_bench(args, dev, rank, world)
    from egm_unet_amd import GRFBUNet
    from egm_unet_amd._lib import lib, require_gpu
    from egm_unet_amd.optim import SGD
    from egm_unet_amd.parallel import GradAllReducer
    from egm_unet_amd.train_utils import criterion
    require_gpu()

    torch.manual_seed(0)                                   # identical initial weights on every rank
    with contextlib.redirect_stdout(sys.stderr):           # the constructor prints a line like the reference's (src/EGM-UNet.py:516); stdout carries the JSON line only
        model = GRFBUNet(3, 2, base_c=32).to(dev).train()
    model.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    opt = SGD(model.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    reducer = GradAllReducer(model, world_size=world) if world > 1 else None
    if reducer is not None:
        opt.grad_scale = 1.0 / world
    x, t = synth_batch(args.batch, args.size, args.size, 1000 + rank, dev)
    lw = torch.tensor([1.0, 2.0], device=dev)

    def eager_step():
        out = model(x)
        loss = criterion(out, t, lw, num_classes=2, ignore_index=255)
        opt.zero_grad()
        loss.backward()
        if reducer is not None:
            opt.grad_source = reducer.finish()
        opt.step()
        return loss

    capture_error = None
    if args.eager:
        step = eager_step
    else:
        from egm_unet_amd.graph import GraphedTrainStep
        try:
            step = GraphedTrainStep(model, opt, x, t, lw, num_classes=2, ignore_index=255, reducer=reducer, warmup=2,
                                    split=False if args.dp_single_graph else None)
        except Exception as e:                                  # capture problems must not cost the measurement: go eager
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
            torch.cuda.synchronize()
            if reducer is not None:
                reducer.hooks_enabled = True
            args.eager = True
            capture_error = f"{type(e).__name__}: {str(e)[:160]}"
            step = eager_step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    coll0 = reducer.collectives_issued if reducer is not None else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enqueue = time.perf_counter() - t0                   # host-side cost of issuing the steps (GPU runs behind)
    coll_per_step = ((reducer.collectives_issued - coll0) / args.steps) if reducer is not None else 0
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss.detach())

    roofline = roofline_wgrad = families = None
    cpu = None
    # Instrumented eager step outside the timed region: every C-ABI call bracketed by HIP events on the launch stream.  At N > 1 it is
    # off unless --instrument (EVERY rank would have to run it, its hook-driven gradient exchange included, after the measurement);
    # EGM_BENCH_NO_INSTRUMENT=1 switches it off at N = 1 too -- the rocprofv3 --pmc passes use that, so that their per-kernel
    # averages cover the graph replays only (tools/pmc_traffic.py).
    instrument = (world == 1 or args.instrument) and not os.environ.get("EGM_BENCH_NO_INSTRUMENT")
    kt = None
    if instrument:
        # after a captured graph the reducer's autograd hooks are off: switch them back on for this eager step
        if reducer is not None:
            reducer.hooks_enabled = True
        from egm_unet_amd import ops as _ops
        grouped = _ops.group_convs()
        _ops.group_convs(False)                 # every conv launched (and timed) on its own: inside a launch group the calls only record
        try:                                    # (the deferred weight-gradient slab kernels stay merged: egm_conv_wgrad_multi is one timed call)
            with KernelTimer(lib()) as kt:
                eager_step()
        finally:
            _ops.group_convs(grouped)
    if rank == 0 and kt is not None:
        agg = kt.summary()
        total_ms = sum(v[1] for v in agg.values())
        dom_key, dom = max(((k, v) for k, v in agg.items() if v[2] > 0), key=lambda kv: kv[1][1])
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command with
        # EGM_BENCH_NO_INSTRUMENT=1, i.e. graph replays only; FETCH_SIZE doubled per the gfx950 note of the guide): a CHECKED-IN
        # measurement keyed by kernel instantiation (tools/pmc_traffic.py)
        import glob
        tpath = next((sorted(g)[-1] for g in (glob.glob(os.path.join(ROOT, "profiles", f"r0{r}*_pmc_traffic.json")) for r in (9, 8, 7, 6, 5, 4)) if g), "")
        tjson = json.load(open(tpath)) if (tpath and args.dtype == "bf16") else {}
        mfma_peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3

        def roofline_of(key, row):
            n_, ms_, flop_, byts_, roof_ms_, n_ung, byts_ung = row
            # the bound is the kernel's own: over ITS launches, which of MFMA time at peak and HBM time at peak is longer
            t_mfma, t_hbm = flop_ / (mfma_peak * 1e12), byts_ / (HBM_PEAK_GBS * 1e9)
            hbm = t_hbm > t_mfma
            if hbm:
                ach, peak, unit = byts_ / (ms_ * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
            else:
                ach, peak, unit = flop_ / (ms_ * 1e-3) / 1e12, mfma_peak, "TFLOP/s"
            tr = tjson.get(key, {}).get("hbm_bytes_per_launch_corrected")
            alg_ung = (byts_ung / n_ung) if n_ung else None
            return {"bound": "hbm" if hbm else "mfma", "achieved": round(ach, 2), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                    "traffic": tr,
                    # like for like: the PMC passes see this kernel name only for the launches that are not merged into a launch group
                    "traffic_over_algorithmic": round(tr / alg_ung, 3) if (tr and alg_ung) else None,
                    "traffic_source": (os.path.relpath(tpath, ROOT) + ": rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over graph replays only "
                                       "(EGM_BENCH_NO_INSTRUMENT=1), per kernel instantiation, recorded earlier and checked in (FETCH doubled per the "
                                       "gfx950 note of MI355X_MICROARCH.md); not collected by this run") if tr else None,
                    "traffic_note": ("the counter bytes include the kernel's fp32 partial-sum slabs (written once, summed by wgrad_reduce_multi_kernel); "
                                     "the algorithmic figure counts the operands x and dy only") if (tr and "wgrad" in key) else None,
                    "kernel": key, "algorithmic_bytes_per_launch": round(byts_ / n_),
                    "algorithmic_bytes_per_launch_in_graph": round(alg_ung) if alg_ung else None, "launches_per_step_in_graph": n_ung,
                    "launches_per_step": n_, "avg_launch_ms": round(ms_ / n_, 4), "algorithmic_gflop_per_launch": round(flop_ / n_ / 1e9, 3),
                    "tflops": round(flop_ / (ms_ * 1e-3) / 1e12, 2), "frac_of_mfma_peak": round(flop_ / (ms_ * 1e-3) / 1e12 / mfma_peak, 4),
                    "share_of_step_kernel_time": round(ms_ / total_ms, 3),
                    # sum over the launches of max(MFMA, HBM)-roofline time / measured time (each launch against its own bound)
                    "frac_of_per_layer_roofline": round(roof_ms_ / ms_, 4) if args.dtype == "bf16" else None}

        roofline = roofline_of(dom_key, dom)
        roofline["note"] = ("dominant = the conv kernel instantiation with the largest share of the step; the step is ~700 launches and no kernel "
                            "exceeds a few percent, see families")
        # the layers the north_star target is quoted on, each against its own roofline
        roofline["encoder_3x3"] = kt.encoder_table() if args.dtype == "bf16" and args.batch == 8 and args.size == 512 else None
        # second object: the heaviest weight-gradient kernel
        wg = [(k, v) for k, v in agg.items() if v[2] > 0 and "wgrad" in k]
        roofline_wgrad = roofline_of(*max(wg, key=lambda kv: kv[1][1])) if wg else None
        families = kt.families()
        top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]
        print(f"[bench] host enqueue {1e3 * t_enqueue / args.steps:.2f} ms/step vs wall {1e3 * elapsed / args.steps:.2f} ms/step", file=sys.stderr)
        print("[bench] kernel time by C-ABI entry (instrumented step, ms): " +
              ", ".join(f"{k}={v[1]:.2f}({v[0]})" for k, v in top) + f"; total {total_ms:.2f}", file=sys.stderr)
        if os.environ.get("EGM_CONV_TABLE"):
            print("\n".join(kt.conv_table()), file=sys.stderr)
    if rank == 0 and not args.no_cpu_baseline:
        if world == 1:
            cpu = cpu_baseline()
        else:
            # timed at N = 1 only (the other ranks would idle at the barrier for its 25 s): carry the newest checked-in N = 1 figure
            for r in (9, 8, 7, 6, 5, 4, 3):
                q = os.path.join(ROOT, "profiles", f"r0{r}_bench_line.json")
                if os.path.exists(q):
                    try:
                        cpu = dict(json.load(open(q))["cpu_baseline"], source=f"n1 ({os.path.relpath(q, ROOT)})")
                    except (KeyError, TypeError, ValueError):
                        cpu = None
                    if cpu:
                        break

    if rank == 0:
        imgs = args.batch * world * args.steps
        line = {
            "metric": "train images/sec at 3x512x512 bs=8/GPU", "value": round(imgs / elapsed, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"EGM-UNet GRFBUNet(3,2,base_c=32) fwd+5-term-loss+bwd+SGD, {args.batch}x3x{args.size}x{args.size} per GPU "
                                   "(BASELINE.json configs[1])",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "launch": (f"eager (capture failed: {capture_error})" if capture_error else "eager") if args.eager else
                       ("hipGraph replay" if world == 1 or not getattr(step, "split", False) else
                        "3 hipGraph replays per step (fwd + decoder bwd | encoder bwd | SGD), RCCL all-reduce of bucket 0 / 1 between them on a side stream"),
                       "allreduces_per_step": coll_per_step, "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "rccl_version": _rccl_version() if world > 1 else None, "dist_backend": (os.environ.get("EGM_DIST_BACKEND", "nccl") + (" (RCCL)" if os.environ.get("EGM_DIST_BACKEND", "nccl") == "nccl" else "")) if world > 1 else None,
                       "final_loss": round(final_loss, 4)},
            "roofline": roofline, "roofline_wgrad": roofline_wgrad, "families": families, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
