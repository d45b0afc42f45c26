#!/bin/bash
# One GPU-box visit: new tests first, then the whole -m gpu suite, then the bench line and a kernel trace of it.
# usage: tools/gpu_round.sh <tag> [pytest-selection for the first, fail-fast pass]
set -o pipefail
tag=${1:-run}; first=${2:-tests/test_gpu_fused_bn.py}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -m pytest $first -x -q -m gpu > $out/first.log 2>&1; rc1=$?
tail -15 $out/first.log
[ $rc1 -ne 0 ] && { echo "first pass failed (rc=$rc1)"; exit $rc1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 1500 $out/bench.json
EGM_FUSE_BN=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_nofuse.json 2> $out/bench_nofuse.err; python - <<PY
import json
for f in ("$out/bench.json", "$out/bench_nofuse.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f, "unreadable", e)
PY
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python tools/prof_summary.py $f 0 $out/kernel_trace_summary.md > /dev/null && head -60 $out/kernel_trace_summary.md
rm -rf $out/prof
python -m pytest tests -q -m gpu -x > $out/all.log 2>&1; echo "full suite rc=$?"; tail -8 $out/all.log
