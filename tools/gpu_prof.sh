#!/bin/bash
# kernel trace of the bench under the given env (comma-separated VAR=VALUE list), full per-kernel table
out=gpurun_out/$1; mkdir -p $out; export TMPDIR=/tmp
for kv in $(echo $2 | tr ',' ' '); do export $kv; done
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
PROF_TOP=400 python tools/prof_summary.py $f 0 $out/kernel_trace_summary.md $out/timeline.tsv > /dev/null; rm -rf $out/prof; head -8 $out/kernel_trace_summary.md
