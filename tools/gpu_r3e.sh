#!/bin/bash
mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3/e_tests.log 2>&1
echo "gpu tests rc=$?" | tee gpurun_out/r3/e_status.log
tail -5 gpurun_out/r3/e_tests.log
grep "bf16 whole-model\|held-out 64\|bf16 vs fp32" gpurun_out/r3/e_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3/e_bench.json 2> gpurun_out/r3/e_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r3/e_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3/e_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["roofline"]["avg_launch_ms"])
for r in d["roofline"]["encoder_3x3"]: print(r)
print(d.get("roofline_wgrad"))
PY
