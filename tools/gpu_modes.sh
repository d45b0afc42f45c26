#!/bin/bash
# A/B of fusion modes on one box: bench line per mode (ms/step), then a kernel trace of the chosen one
out=gpurun_out/$1; mkdir -p $out; shift
export TMPDIR=/tmp
for mode in "$@"; do
  env $(echo $mode | tr ',' ' ') python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$mode.json 2> $out/bench_$mode.err
  python - <<PY
import json
try:
    d = json.loads(open("$out/bench_$mode.json").read().strip().splitlines()[-1]); print("$mode", d["ms_per_step"], d["value"])
except Exception as e: print("$mode", "unreadable", e)
PY
done
