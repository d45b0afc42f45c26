#!/bin/bash
# A/B of bench.py under different environments in ONE box visit, interleaved twice: tools/gpu_ab.sh "VAR=a" "VAR=b" ...
export TMPDIR=/tmp; mkdir -p gpurun_out/ab
set -euo pipefail
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab/out.json 2> gpurun_out/ab/err.txt
    python -c "
import json; d=json.loads(open('gpurun_out/ab/out.json').read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'])"
  done
done
