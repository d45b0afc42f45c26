#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 600 python tools/conv_tile_diag.py > gpurun_out/r3/b_diag_ablate.log 2>&1
echo "ablate rc=$?" | tee -a gpurun_out/r3/b_status.log
cat gpurun_out/r3/b_diag_ablate.log
EGM_LIB_TAG=timing timeout -k 10 600 python tools/conv_tile_diag.py > gpurun_out/r3/b_diag_timing.log 2>&1
echo "timing rc=$?" | tee -a gpurun_out/r3/b_status.log
cat gpurun_out/r3/b_diag_timing.log
timeout -k 10 900 python -m pytest tests/test_gpu_parallel.py -x -q -k "rccl or starts_its_own or tensor_hook" > gpurun_out/r3/b_parallel.log 2>&1
echo "parallel tests rc=$?" | tee -a gpurun_out/r3/b_status.log
tail -15 gpurun_out/r3/b_parallel.log
timeout -k 10 900 python -m pytest tests/test_gpu_egm.py -x -q -s -k "bf16_gradients" > gpurun_out/r3/b_bf16grad.log 2>&1
echo "bf16 grad rc=$?" | tee -a gpurun_out/r3/b_status.log
grep "bf16 whole-model\|passed\|failed" gpurun_out/r3/b_bf16grad.log
