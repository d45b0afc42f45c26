"""CPU: the C-ABI shared library builds, loads, and exports every symbol include/egm_hip.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from egm_unet_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built_lib):
    from egm_unet_amd._lib import parse_header
    protos = parse_header()
    assert len(protos) >= 30
    cdll = ctypes.CDLL(built_lib)
    missing = [n for n in protos if not hasattr(cdll, n)]
    assert not missing, missing
    cdll.egm_version.restype = ctypes.c_int
    assert cdll.egm_version() >= 100


def test_header_is_plain_c(tmp_path):
    """The boundary header must compile as C (no C++/torch types in the signatures)."""
    src = tmp_path / "t.c"
    src.write_text('#include "egm_hip.h"\nint main(void) { int (*f)(void) = egm_version; return f == 0 ? 1 : 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_argument_validation_without_gpu(built_lib):
    """Host-side shape checks run before any launch: bad arguments return EGM_ERR_ARG with a message."""
    from egm_unet_amd._lib import lib
    L = lib()
    rc = L.cdll.egm_conv_fwd(1, None, 8, None, None, 0, None, 8, None, 1, 8, 8, 8, 8, 3, 3, 1, None)
    assert rc == -1 and b"null pointer" in L.cdll.egm_last_error()
    assert L.cdll.egm_conv_stats_tiles(0, 8, 512, 512, 32, 32, 3, 3, 1) == 8 * 64 * 16
    assert L.cdll.egm_conv_stats_tiles(1, 8, 512, 512, 64, 32, 3, 3, 1) == 512      # 64 -> 32: 4-wave kernel, two pixel groups per CU
    assert L.cdll.egm_conv_stats_tiles(1, 8, 512, 512, 32, 32, 3, 3, 1) == 256      # 32 -> 32: weights-in-registers kernel, one per CU
    assert L.cdll.egm_conv_stats_tiles(1, 8, 256, 256, 64, 64, 3, 3, 1) == 256      # 8-wave tile kernel: one pixel group per CU
    assert L.cdll.egm_conv_tile_mode(0) == 5
    assert L.cdll.egm_conv_stats_tiles(1, 8, 256, 256, 64, 64, 3, 3, 1) == 512      # switched off: the 4-wave kernel again
    assert L.cdll.egm_conv_stats_tiles(1, 8, 512, 512, 32, 32, 3, 3, 1) == 512
    assert L.cdll.egm_conv_tile_mode(5) == 0
    assert L.cdll.egm_loss_workspace(8, 2) > 0
    assert L.cdll.egm_conv_wgrad_workspace(8, 512, 512, 64, 32, 3, 3) > 0


def test_product_never_imports_oracle():
    """The product package must not reach into oracle/ (parity claims depend on it)."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "egm_unet_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                if "import oracle" in txt or "from oracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import egm_unet_amd._lib as L
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L._Lib()


def test_model_refuses_cpu_input():
    import torch
    from egm_unet_amd import UNet
    m = UNet(3, 2, base_c=8)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 16, 16))
