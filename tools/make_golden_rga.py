#!/usr/bin/env python3
"""Generate tests/golden/rga_d128_o3.npz: the reference's RecursiveGatedAttention(128, order=3) (src/EGM-UNet.py:458-547), forward
+ backward, through tools/make_golden.py's loader and block_fixture.  Build container only.  Re-run: python tools/make_golden_rga.py"""
import os
import sys

import torch

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402


def main():
    ref = G.load_reference()
    egm = ref[1] if isinstance(ref, (tuple, list)) else ref
    if not hasattr(egm, "RecursiveGatedAttention"):
        egm = [m for m in (ref if isinstance(ref, (tuple, list)) else [ref]) if hasattr(m, "RecursiveGatedAttention")][0]
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    m = egm.RecursiveGatedAttention(128, order=3)
    with torch.no_grad():
        m.scale.fill_(1.1)
    G.block_fixture("rga_d128_o3", m, [torch.randn(2, 128, 8, 12, generator=g)])


if __name__ == "__main__":
    main()
