"""Build libegm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m egm_unet_amd.build            # incremental
    python -m egm_unet_amd.build --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
TAG = os.environ.get("EGM_BUILD_TAG", "")          # diagnostic builds (with EGM_HIPCC_EXTRA) live beside the product library
LIB = os.path.join(LIBDIR, "libegm_hip%s.so" % ("_" + TAG if TAG else ""))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("EGM_HIPCC_EXTRA", "").split()        # e.g. -DEGM_CONV_TIMING for tools/diag_conv_phases.py


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build" + ("_" + TAG if TAG else ""))
    os.makedirs(objdir, exist_ok=True)
    # objects and the library are only reusable under the flags they were built with (a library left over from a diagnostic
    # build, e.g. -DEGM_CONV_TIMING, must never be picked up by a normal one): the flag set is stamped next to the objects
    stamp_path = os.path.join(LIBDIR, "flags%s.stamp" % ("_" + TAG if TAG else ""))          # next to the library, so it travels with it
    stamp = " ".join([HIPCC] + FLAGS)
    if not os.path.exists(stamp_path) or open(stamp_path).read() != stamp:
        force = True
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "egm_hip.h"))
    jobs = []
    for src in _sources():
        obj = os.path.join(objdir, src[:-4] + ".o")
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    failed = False
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(compile_one, jobs):
            if verbose and (rc != 0 or out.strip()):
                print(f"[hipcc {src}] rc={rc}\n{out}", file=sys.stderr)
            failed |= rc != 0
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    objs = [os.path.join(objdir, s[:-4] + ".o") for s in _sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"built {LIB} ({len(jobs)} recompiled)")
    with open(stamp_path, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
