#!/bin/bash
# Round-end evidence: kernel trace + stats of the bench command, PMC passes (traffic, MFMA busy), ablation table, bench line
tag=${1:-r03}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 600 $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1); st=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && PROF_TOP=60 python tools/prof_summary.py $f 0 $out/kernel_trace_summary.md $out/timeline.tsv > /dev/null && head -12 $out/kernel_trace_summary.md
[ -n "$st" ] && cp $st $out/kernel_stats.csv
rm -rf $out/prof
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -o f -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -o w -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/pmc_w.log 2>&1
ff=$(find $out/pf -name "*counter_collection.csv" | head -1); fw=$(find $out/pw -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py $ff $fw $out/pmc_traffic.json > /dev/null && echo traffic ok
rm -rf $out/pf $out/pw
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pm -o m -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/pmc_m.log 2>&1
fm=$(find $out/pm -name "*counter_collection.csv" | head -1)
[ -n "$fm" ] && python tools/pmc_mfma.py $fm $out/pmc_mfma_util.json
rm -rf $out/pm
python bench.py --workload ablation_1024 --steps 10 --warmup 3 > $out/ablation_1024.json 2> $out/ablation.err; echo "ablation rc=$?"
