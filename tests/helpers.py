"""Shared helpers for the parity tests (fixtures -> oracle state dicts, comparisons)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def fixture_state(fx, prefix="m", group="state", requires_grad=True):
    """Arrays stored as '<group>/<key>' -> {'<prefix>.<key>': tensor}."""
    st = {}
    for k, v in fx.items():
        if k.startswith(group + "/"):
            t = torch.from_numpy(np.array(v))
            name = (prefix + "." if prefix else "") + k[len(group) + 1:]
            if requires_grad and t.is_floating_point() and "running_" not in name:
                t.requires_grad_(True)
            st[name] = t
    return st


def rel_err(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def assert_close(a, b, rtol, atol, what=""):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; "
                           f"max abs err {float(err.max()):.3e}, rel-L2 {rel_err(a, b):.3e}")
