#!/bin/bash
mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3/i_tests.log 2>&1
echo "gpu tests rc=$?" | tee gpurun_out/r3/i_status.log
tail -4 gpurun_out/r3/i_tests.log
grep "bf16 whole-model\|held-out 64\|bf16 vs fp32" gpurun_out/r3/i_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3/i_bench.json 2> gpurun_out/r3/i_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3/i_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
for r in d["roofline"]["encoder_3x3"]: print(r["layer"], r["us"], r["frac_of_layer_roofline"])
PY
python bench.py --workload clipseg_infer --steps 10 --warmup 3 > gpurun_out/r3/i_clipseg.json 2> gpurun_out/r3/i_clipseg.err; echo "clipseg rc=$?"; cat gpurun_out/r3/i_clipseg.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/i_cs -o p -- python bench.py --workload clipseg_infer --steps 6 --warmup 2 > gpurun_out/r3/i_cs.log 2>&1
f=$(find gpurun_out/r3/i_cs -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && PROF_TOP=30 python tools/prof_summary.py $f 8 gpurun_out/r3/i_clipseg_trace_summary.md > /dev/null && head -30 gpurun_out/r3/i_clipseg_trace_summary.md
rm -rf gpurun_out/r3/i_cs
