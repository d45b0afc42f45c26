"""ctypes binding of libegm_hip.so (the C ABI declared in include/egm_hip.h).

The prototypes are parsed from the header itself, so the header is the single
source of truth for the boundary.  There is NO fallback: if the shared library
is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "egm_hip.h")
# EGM_LIB_TAG selects a diagnostic build made with `EGM_BUILD_TAG=<tag> python -m egm_unet_amd.build` (lib/libegm_hip_<tag>.so)
LIB_PATH = os.path.join(_HERE, "lib", "libegm_hip%s.so" % ("_" + os.environ["EGM_LIB_TAG"] if os.environ.get("EGM_LIB_TAG") else ""))

EGM_F32, EGM_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU = 0, 1, 2, 3

_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong,
    "egm_stream_t": ctypes.c_void_p, "unsigned long long": ctypes.c_ulonglong,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|long long|const char\*|void)\s+(egm_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "const char*": ctypes.c_char_p, "void": None}[ret]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    t = re.sub(r"\bconst\b", "", a).strip()
                    t = " ".join(t.split()[:-1])        # drop the parameter name
                    argtypes.append(_CTYPES[t])
        protos[name] = (restype, argtypes)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
                "Build it with `python -m egm_unet_amd.build` (hipcc, --offload-arch=gfx950).")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self.cdll, name, None)
            if fn is None:
                raise RuntimeError(f"libegm_hip.so does not export {name} (declared in include/egm_hip.h)")
            fn.restype = restype
            fn.argtypes = argtypes

    def call(self, name, *args):
        """Call an int-returning entry point; raise on a non-zero status."""
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {self.cdll.egm_last_error().decode()}")

    def query(self, name, *args):
        """Call a size/count query (returns the value; negative = error)."""
        v = getattr(self.cdll, name)(*args)
        if v < 0:
            raise RuntimeError(f"{name} failed ({v}): {self.cdll.egm_last_error().decode()}")
        return v


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


_gpu_ok = False


def require_gpu():
    """Fail loudly unless a gfx950 device is usable through the HIP library (checked once per process)."""
    global _gpu_ok
    if _gpu_ok:
        return
    if not torch.cuda.is_available():
        raise RuntimeError("egm_unet_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False")
    if not lib().cdll.egm_device_ok():
        raise RuntimeError("egm_unet_amd: " + lib().cdll.egm_last_error().decode())
    _gpu_ok = True


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def dtype_code(dt):
    if dt == torch.float32:
        return EGM_F32
    if dt == torch.bfloat16:
        return EGM_BF16
    raise RuntimeError(f"egm_unet_amd: unsupported activation dtype {dt} (float32 or bfloat16)")
