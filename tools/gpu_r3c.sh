#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_fwd_bwd" > gpurun_out/r3/c_conv_tests.log 2>&1
echo "conv tests rc=$?" | tee gpurun_out/r3/c_status.log
tail -3 gpurun_out/r3/c_conv_tests.log
timeout -k 10 600 python tools/conv_tile_diag.py > gpurun_out/r3/c_diag_ablate.log 2>&1
echo "ablate rc=$?" | tee -a gpurun_out/r3/c_status.log
cat gpurun_out/r3/c_diag_ablate.log
EGM_LIB_TAG=timing timeout -k 10 600 python tools/conv_tile_diag.py > gpurun_out/r3/c_diag_timing.log 2>&1
echo "timing rc=$?" | tee -a gpurun_out/r3/c_status.log
cat gpurun_out/r3/c_diag_timing.log
timeout -k 10 600 python tools/conv_tile_bench.py 5 20 > gpurun_out/r3/c_tile_bench.log 2>&1
echo "tile bench rc=$?" | tee -a gpurun_out/r3/c_status.log
python - <<'PY'
import json
for ln in open("gpurun_out/r3/c_tile_bench.log"):
    if ln.startswith("{"):
        r = json.loads(ln); print(f'{r["layer"]:18s} {r["shape"]:14s} old {r["old_us"]:6.1f} new {r["new_us"]:6.1f}  frac {r["old_frac"]:.3f} -> {r["new_frac"]:.3f}  {r["kernel"][20:]} diff {r["maxdiff"]}')
    elif ln.startswith("total"): print(ln)
PY
