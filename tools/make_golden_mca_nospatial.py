#!/usr/bin/env python3
"""tests/golden/mca_nospatial_c*.npz: the REFERENCE's MCALayer(inp, no_spatial=True) (src/EGM-UNet.py:686-791: two gates, x_out =
(x_h + x_w) / 2, no c_hw parameters), forward + backward, through the same block_fixture as tools/make_golden.py.  Build container only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import block_fixture, load_reference

egm = load_reference()[0]
g = torch.Generator().manual_seed(77)
for c, hw in ((64, (12, 20)), (16, (17, 9))):
    torch.manual_seed(c)
    m = egm.MCALayer(c, no_spatial=True)
    assert not hasattr(m, "c_hw")
    x = torch.relu(torch.randn(2, c, *hw, generator=g))
    block_fixture(f"mca_nospatial_c{c}", m, [x])
