#!/bin/bash
mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -k "upcat or unet or Up or fixture" > gpurun_out/r3/j_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r3/j_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/j_bench.json 2> gpurun_out/r3/j_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r3/j_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3/j_prof -o p -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3/j_prof.log 2>&1
f=$(find gpurun_out/r3/j_prof -name "*kernel_trace.csv" | head -1)
PROF_TOP=80 python tools/prof_summary.py $f 0 gpurun_out/r3/j_summary.md gpurun_out/r3/j_timeline.tsv > /dev/null; rm -rf gpurun_out/r3/j_prof
grep "upcat\|total kernel" gpurun_out/r3/j_summary.md | head
