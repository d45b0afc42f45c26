#!/bin/bash
# Evidence run on the GPU box, one script for every round:  tools/gpu_final.sh <tag> [steps ...]
#   steps (default: all, in this order): tests bench trace pmc ablation
#   tests     python -m pytest tests -m gpu (optionally narrowed by PYTEST_ARGS)
#   bench     the headline line -> $out/bench.json
#   trace     rocprofv3 --kernel-trace --stats of the bench command -> kernel_trace_summary.md, timeline.tsv, kernel_stats.csv
#   pmc       three separate --pmc passes over GRAPH REPLAYS ONLY (EGM_BENCH_NO_INSTRUMENT=1): FETCH_SIZE, WRITE_SIZE -> pmc_traffic.json;
#             SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE -> pmc_mfma_util.json.  The raw counter CSVs are kept (gzip) beside them.
#   ablation  bench.py --workload ablation_1024 -> ablation_1024.json
# A failing step stops the script (no bench / profile numbers beside a red test run); every GPU step has its own timeout.
set -euo pipefail
tag=${1:?usage: tools/gpu_final.sh <tag> [tests bench trace pmc ablation]}; shift || true
steps=${*:-tests bench trace pmc ablation}
out=gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
has() { [[ " $steps " == *" $1 "* ]]; }

if has tests; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > "$out/tests.log" 2>&1 || { tail -40 "$out/tests.log"; echo "tests FAILED"; exit 1; }
  tail -2 "$out/tests.log"
fi
if has bench; then
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err" || { tail -30 "$out/bench.err"; exit 1; }
  python -c "import json,sys; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['bound'], d['roofline']['frac'])"
fi
if has trace; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o p -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline > "$out/prof.log" 2>&1
  f=$(find "$out/prof" -name "*kernel_trace.csv" | head -1); st=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
  PROF_TOP=70 python tools/prof_summary.py "$f" 0 "$out/kernel_trace_summary.md" "$out/timeline.tsv" > /dev/null
  head -14 "$out/kernel_trace_summary.md"
  cp "$st" "$out/kernel_stats.csv"
  rm -rf "$out/prof"
fi
if has pmc; then
  export EGM_BENCH_NO_INSTRUMENT=1
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pf" -o f -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > "$out/pmc_f.log" 2>&1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pw" -o w -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > "$out/pmc_w.log" 2>&1
  ff=$(find "$out/pf" -name "*counter_collection.csv" | head -1); fw=$(find "$out/pw" -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$ff" "$fw" "$out/pmc_traffic.json" > /dev/null && echo "traffic ok"
  python tools/pmc_compact.py "$ff" "$out/pmc_fetch_per_kernel.csv.gz"; python tools/pmc_compact.py "$fw" "$out/pmc_write_per_kernel.csv.gz"
  rm -rf "$out/pf" "$out/pw"
  timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pm" -o m -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > "$out/pmc_m.log" 2>&1
  fm=$(find "$out/pm" -name "*counter_collection.csv" | head -1)
  python tools/pmc_mfma.py "$fm" "$out/pmc_mfma_util.json"
  python tools/pmc_compact.py "$fm" "$out/pmc_mfma_per_kernel.csv.gz"
  rm -rf "$out/pm"
  unset EGM_BENCH_NO_INSTRUMENT
fi
if has ablation; then
  timeout -k 10 500 python bench.py --workload ablation_1024 --steps 10 --warmup 3 > "$out/ablation_1024.json" 2> "$out/ablation.err" || { tail -30 "$out/ablation.err"; exit 1; }
  grep "EdgeEnhancedGRFB\|MCALayer" "$out/ablation.err" | head -4
fi
echo "evidence $tag: done ($steps)"
