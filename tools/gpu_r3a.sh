#!/bin/bash
# round-3 GPU visit A: parity of the new 3x3 kernel, A/B timing, the parallel tests
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_fwd_bwd" > gpurun_out/r3/a_conv_tests.log 2>&1
echo "conv tests rc=$?" | tee -a gpurun_out/r3/a_status.log
tail -5 gpurun_out/r3/a_conv_tests.log
timeout -k 10 600 python tools/conv_tile_bench.py 5 20 > gpurun_out/r3/a_tile_bench.log 2>&1
echo "tile bench rc=$?" | tee -a gpurun_out/r3/a_status.log
cat gpurun_out/r3/a_tile_bench.log | cut -c1-400
timeout -k 10 1200 python -m pytest tests/test_gpu_parallel.py -x -q > gpurun_out/r3/a_parallel.log 2>&1
echo "parallel tests rc=$?" | tee -a gpurun_out/r3/a_status.log
tail -15 gpurun_out/r3/a_parallel.log
