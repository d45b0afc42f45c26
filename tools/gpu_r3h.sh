#!/bin/bash
mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "tile_kernel" > gpurun_out/r3/h_tests.log 2>&1
echo "tile tests rc=$?"; tail -4 gpurun_out/r3/h_tests.log
NEW_MODE=5 ONLY=32x32 timeout -k 10 500 python tools/conv_tile_bench.py 7 20 > gpurun_out/r3/h_bench.log 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r3/h_bench.log"):
    if ln.startswith("{"):
        r = json.loads(ln); print(f'{r["layer"]:18s} {r["shape"]:14s} old {r["old_us"]:6.1f} new {r["new_us"]:6.1f}  frac {r["old_frac"]:.3f} -> {r["new_frac"]:.3f}  {r["kernel"][:40]} diff {r["maxdiff"]}')
    elif ln.startswith("total"): print(ln.strip())
PY
NEW_MODE=5 EGM_LIB_TAG=timing python tools/conv_tile_diag.py 8 512 512 32 32 2>&1 | grep -v amdgpu.ids
