#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation on CPU.

Runs only in the build container (needs /root/reference).  The reference is
imported by file path (its src/__init__.py is broken and EGM-UNet.py is not an
importable name) with an inert stand-in for the unused `thop` import
(src/EGM-UNet.py:6).  Only DATA is written: seeded inputs, the reference
modules' parameters as arrays, outputs and gradients.  No reference source is
copied.  Re-run:  python tools/make_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_reference():
    thop = types.ModuleType("thop")
    thop.profile = lambda *a, **k: (0, 0)
    sys.modules["thop"] = thop
    egm = _load("ref_egm_unet", f"{REF}/src/EGM-UNet.py")
    unet = _load("ref_unet", f"{REF}/src/unet.py")
    sys.path.insert(0, REF)
    from train_utils import train_and_eval, dice_coefficient_loss, distributed_utils
    return egm, unet, train_and_eval, dice_coefficient_loss, distributed_utils


def randomize_bn(module, gen):
    """Non-trivial BN affine so gamma/beta paths are exercised."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))


def block_fixture(name, module, inputs, train=True, extra=None):
    """Forward+backward of a reference module; saves params (pre-step), inputs, output, grads, post-fwd buffers."""
    module.train(train)
    pre = {k: v.detach().clone() for k, v in module.state_dict().items()}
    xs = [x.clone().requires_grad_(True) for x in inputs]
    out = module(*xs)
    if isinstance(out, dict):
        out = out["out"]
    gen = torch.Generator().manual_seed(1234)
    gout = torch.randn(out.shape, generator=gen)
    (out * gout).sum().backward()
    d = {}
    for k, v in pre.items():
        d["state/" + k] = v.numpy()
    for k, v in module.state_dict().items():
        if "running_" in k:
            d["post/" + k] = v.detach().numpy()
    for i, x in enumerate(xs):
        d[f"in{i}"] = x.detach().numpy()
        d[f"gin{i}"] = x.grad.numpy()
    d["out"] = out.detach().numpy()
    d["gout"] = gout.numpy()
    for k, p in module.named_parameters():
        if p.grad is not None:
            d["grad/" + k] = p.grad.numpy()
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"{name}: out {tuple(out.shape)} mean {out.mean().item():+.5f}  ({len(d)} arrays)")


def synth_target(n, h, w, gen, ignore=True):
    t = torch.zeros(n, h, w, dtype=torch.int64)
    for i in range(n):
        y0, x0 = torch.randint(0, h // 2, (2,), generator=gen).tolist()
        t[i, y0:y0 + h // 3, x0:x0 + w // 2] = 1
    if ignore:
        t[torch.rand(n, h, w, generator=gen) < 0.02] = 255
    return t


def main():
    os.makedirs(OUT, exist_ok=True)
    egm, unet, tae, dcl, du = load_reference()
    g = torch.Generator().manual_seed(0)

    # ---- manifest: state_dict keys/shapes + seeded-init checksums (drop-in boundary) ----
    manifest = {}
    for tag, ctor in (("egm_unet_3_2_32", lambda: egm.GRFBUNet(3, 2, base_c=32)),
                      ("unet_default", lambda: unet.UNet()),
                      ("egm_unet_3_2_8", lambda: egm.GRFBUNet(3, 2, base_c=8))):
        torch.manual_seed(0)
        m = ctor()
        sd = m.state_dict()
        manifest[tag] = {
            "n_entries": len(sd),
            "n_params": sum(p.numel() for p in m.parameters()),
            "keys": {k: list(v.shape) for k, v in sd.items()},
            "init_sum": {k: float(v.double().sum()) for k, v in sd.items() if v.is_floating_point()},
        }
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f)
    print("manifest:", {k: v["n_entries"] for k, v in manifest.items()})

    # ---- per-block fixtures ----
    torch.manual_seed(1)
    m = unet.DoubleConv(8, 16); randomize_bn(m, g)
    block_fixture("double_conv", m, [torch.randn(2, 8, 20, 24, generator=g)])
    m = unet.Up(32, 8, bilinear=True); randomize_bn(m, g)
    block_fixture("up_block", m, [torch.randn(2, 16, 10, 12, generator=g), torch.randn(2, 16, 20, 24, generator=g)])
    m = unet.Up(32, 8, bilinear=True); randomize_bn(m, g)   # odd skip size -> exercises the zero pad
    block_fixture("up_block_pad", m, [torch.randn(1, 16, 7, 9, generator=g), torch.randn(1, 16, 15, 19, generator=g)])

    for c, hw in ((64, (12, 20)), (256, (8, 8)), (16, (16, 16))):
        m = egm.MCALayer(c)
        x = torch.relu(torch.randn(2, c, *hw, generator=g))      # post-ReLU like in the network
        block_fixture(f"mca_c{c}", m, [x])

    m = egm.EdgeAwareFeatureEnhancer(16); randomize_bn(m, g)
    block_fixture("edge_gate", m, [torch.randn(2, 16, 12, 14, generator=g)])

    m = egm.FusionConv(28, 16); block_fixture("fusion_conv", m, [torch.randn(2, 28, 12, 14, generator=g)] * 1 + [torch.randn(2, 28, 12, 14, generator=g)])

    m = egm.EdgeEnhancedGRFB(64, 64, stride=1, scale=0.1, visual=12); randomize_bn(m, g)
    block_fixture("edge_grfb_c64", m, [torch.relu(torch.randn(2, 64, 40, 44, generator=g))])
    m = egm.EdgeEnhancedGRFB(32, 32, stride=1, scale=0.1, visual=12); randomize_bn(m, g)
    block_fixture("edge_grfb_c32", m, [torch.relu(torch.randn(1, 32, 16, 16, generator=g))])

    m = egm.RecursiveGatedAttention(64)
    with torch.no_grad():
        m.scale.fill_(0.9)
    block_fixture("rga_d64", m, [torch.randn(2, 64, 8, 8, generator=g)])

    m = egm.Down(8, 16); randomize_bn(m, g)
    block_fixture("egm_down", m, [torch.randn(2, 8, 32, 32, generator=g)])

    # ---- whole models (small) ----
    torch.manual_seed(2)
    m = unet.UNet(3, 2, base_c=8); randomize_bn(m, g)
    block_fixture("unet_b8", m, [torch.randn(2, 3, 64, 64, generator=g)])
    m = egm.GRFBUNet(3, 2, base_c=8); randomize_bn(m, g)
    block_fixture("egm_unet_b8", m, [torch.randn(2, 3, 64, 64, generator=g)])
    # ablation twin without MCALayer (src/yuanGRFBUNet.py)
    yuan = _load("ref_yuan_unet", f"{REF}/src/yuanGRFBUNet.py")
    my = yuan.GRFBUNet(3, 2, base_c=8); randomize_bn(my, g)
    block_fixture("yuan_unet_b8", my, [torch.randn(2, 3, 64, 64, generator=g)])
    # eval-mode forward of the same EGM-UNet after one train-mode forward (running stats in use)
    m.eval()
    with torch.no_grad():
        xe = torch.randn(2, 3, 64, 64, generator=g)
        ye = m(xe)["out"]
    np.savez_compressed(os.path.join(OUT, "egm_unet_b8_eval.npz"), x=xe.numpy(), out=ye.numpy(),
                        **{"state/" + k: v.numpy() for k, v in m.state_dict().items()})

    # ---- criterion / metrics ----
    for tag, (n, h, w) in (("crit_small", (2, 32, 32)), ("crit_mid", (3, 40, 56))):
        logits = torch.randn(n, 2, h, w, generator=g) * 2
        target = synth_target(n, h, w, g)
        if tag == "crit_small":
            target[1][target[1] != 255] = 0          # an all-background sample
        lw = torch.tensor([1.0, 2.0])
        x = logits.clone().requires_grad_(True)
        loss = tae.criterion({"out": x}, target, lw, num_classes=2, ignore_index=255)
        loss.backward()
        terms = {
            "ce": torch.nn.functional.cross_entropy(logits, target, ignore_index=255, weight=lw),
            "dice": dcl.dice_loss(logits, dcl.build_target(target, 2, 255), multiclass=True, ignore_index=255),
            "laplace": dcl.laplace_loss(logits), "lap": dcl.lap_loss(logits, target), "sobel": dcl.sobel_loss(logits, target),
        }
        cm = du.ConfusionMatrix(2); cm.update(target.flatten(), logits.argmax(1).flatten())
        acc_g, acc, iu = cm.compute()
        dc = du.DiceCoefficient(num_classes=2, ignore_index=255); dc.update(logits, target)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), logits=logits.numpy(), target=target.numpy(),
                            loss=loss.detach().numpy(), grad=x.grad.numpy(),
                            confmat=cm.mat.numpy(), acc_global=acc_g.numpy(), acc=acc.numpy(), iu=iu.numpy(),
                            dice_metric=dc.value.numpy(), **{"term_" + k: v.numpy() for k, v in terms.items()})
        print(tag, float(loss), {k: round(float(v), 5) for k, v in terms.items()})
    # no-dice branch and the all-ignored-free default (ignore_index=-100)
    logits = torch.randn(2, 2, 16, 16, generator=g); target = synth_target(2, 16, 16, g, ignore=False)
    np.savez_compressed(os.path.join(OUT, "crit_noignore.npz"), logits=logits.numpy(), target=target.numpy(),
                        loss=tae.criterion({"out": logits}, target).numpy(),
                        loss_nodice=tae.criterion({"out": logits}, target, dice=False).numpy())

    # ---- config 1 plumbing value (BASELINE.json configs[0]): UNet() defaults, seed 0 ----
    torch.manual_seed(0)
    m = unet.UNet()
    x = torch.randn(1, 1, 256, 256); t = torch.randint(0, 2, (1, 256, 256))
    m.train()
    out = m(x)["out"]
    dl = dcl.dice_loss(out, dcl.build_target(t, 2, 255), multiclass=True, ignore_index=255)
    tot = tae.criterion({"out": out}, t, None, num_classes=2, ignore_index=255)
    np.savez_compressed(os.path.join(OUT, "config1_unet_default.npz"), dice_loss=dl.detach().numpy(),
                        criterion=tot.detach().numpy(), out_mean=out.mean().detach().numpy(),
                        out_abs_mean=out.abs().mean().detach().numpy(),
                        out_crop=out[0, :, 100:116, 100:116].detach().numpy())
    print("config1: dice", float(dl), "criterion", float(tot))

    # ---- 3-step training trace of EGM-UNet(base_c=8): loss + probe weights (train_one_epoch semantics) ----
    torch.manual_seed(3)
    m = egm.GRFBUNet(3, 2, base_c=8)
    init = {k: v.detach().clone() for k, v in m.state_dict().items()}
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.02, momentum=0.9, weight_decay=1e-4)
    xs = torch.randn(3, 2, 3, 64, 64, generator=g)
    ts = torch.stack([synth_target(2, 64, 64, g) for _ in range(3)])
    lw = torch.tensor([1.0, 2.0])
    losses = []
    m.train()
    for s in range(3):
        loss = tae.criterion(m(xs[s]), ts[s], lw, num_classes=2, ignore_index=255)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss))
    probes = ["in_conv.0.weight", "down1.1.3.c_hw.weight", "down2.1.7.fusion_conv.down.weight", "attn1.scale",
              "up4.conv.3.weight", "out_conv.0.bias", "down1.1.1.running_mean", "down3.1.7.shortcut.bn.running_var"]
    final = m.state_dict()
    np.savez_compressed(os.path.join(OUT, "train3_egm_b8.npz"), xs=xs.numpy(), ts=ts.numpy(), losses=np.array(losses),
                        **{"init/" + k: v.numpy() for k, v in init.items()},
                        **{"final/" + k: final[k].detach().numpy() for k in probes})
    print("train3 losses", losses)

    # ---- LR schedule ----
    sched_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.02)
    sch = tae.create_lr_scheduler(sched_opt, num_step=7, epochs=5, warmup=True)
    lrs = []
    for _ in range(35):
        lrs.append(sched_opt.param_groups[0]["lr"]); sched_opt.step(); sch.step()
    np.savez_compressed(os.path.join(OUT, "lr_schedule.npz"), lrs=np.array(lrs))

    # ---- block-level ablation twin: the plain GRFB (own generator, so the fixtures above keep their values) ----
    g2 = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    m = egm.GRFB(64, 64, stride=1, scale=0.1, visual=12); randomize_bn(m, g2)
    block_fixture("plain_grfb_c64", m, [torch.relu(torch.randn(2, 64, 40, 44, generator=g2))])
    m = egm.ELA(64, kernel_size=7)
    with torch.no_grad():
        m.gn.weight.copy_(1.0 + 0.2 * torch.randn(64, generator=g2)); m.gn.bias.copy_(0.2 * torch.randn(64, generator=g2))
    block_fixture("ela_c64", m, [torch.randn(2, 64, 24, 36, generator=g2) + 0.3])
    m = egm.ELA(32, kernel_size=5)
    block_fixture("ela_c32_k5", m, [torch.randn(1, 32, 17, 9, generator=g2)])
    # Up(bilinear=False): ConvTranspose2d(2, stride 2) + pad + cat + DoubleConv (src/unet.py:35-51); even and odd skip sizes
    torch.manual_seed(79)
    m = unet.Up(32, 16, bilinear=False); randomize_bn(m, g2)
    block_fixture("up_block_convT", m, [torch.randn(2, 32, 10, 12, generator=g2), torch.randn(2, 16, 20, 24, generator=g2)])
    m = unet.Up(32, 16, bilinear=False); randomize_bn(m, g2)
    block_fixture("up_block_convT_pad", m, [torch.randn(1, 32, 7, 9, generator=g2), torch.randn(1, 16, 15, 19, generator=g2)])
    mm = unet.UNet(3, 2, bilinear=False, base_c=8)
    json.dump({k: list(v.shape) for k, v in mm.state_dict().items()}, open(os.path.join(OUT, "unet_convT_manifest.json"), "w"))
    torch.manual_seed(78)
    m = egm.HEGDC(16, 24); randomize_bn(m, g2)
    with torch.no_grad():
        m.alpha.fill_(0.8); m.den.fill_(0.3)
    block_fixture("hegdc_16_24", m, [torch.randn(2, 16, 20, 28, generator=g2)])


if __name__ == "__main__":
    main()
