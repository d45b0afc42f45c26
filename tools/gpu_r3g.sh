#!/bin/bash
mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_fwd_bwd" > gpurun_out/r3/g_conv_tests.log 2>&1
echo "conv tests rc=$?"; tail -3 gpurun_out/r3/g_conv_tests.log
for v in 0 1; do
  EGM_WGRAD8=$v EGM_CONV_TABLE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/g_bench_w8_$v.json 2> gpurun_out/r3/g_bench_w8_$v.err
  echo "== EGM_WGRAD8=$v"; python -c "
import json,sys
d=json.loads(open('gpurun_out/r3/g_bench_w8_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  grep "^wgrad     k3 d1" gpurun_out/r3/g_bench_w8_$v.err | awk '1' | head -24
done
