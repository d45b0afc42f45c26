#!/bin/bash
# visit D: product vs dma2 build A/B, true kernel durations (rocprofv3 kernel trace), SQ counters of the 128->128 layer
mkdir -p gpurun_out/r3; export TMPDIR=/tmp
bench() { python - "$1" <<'PY'
import json, sys
tot0 = tot1 = 0
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        r = json.loads(ln); print(f'{r["layer"]:18s} {r["shape"]:14s} old {r["old_us"]:6.1f} new {r["new_us"]:6.1f}  frac {r["old_frac"]:.3f} -> {r["new_frac"]:.3f}  {r["kernel"][20:]}')
    elif ln.startswith("total"): print(ln.strip())
PY
}
timeout -k 10 500 python tools/conv_tile_bench.py 5 20 > gpurun_out/r3/d_bench_prod.log 2>&1; echo "== product build"; bench gpurun_out/r3/d_bench_prod.log
EGM_LIB_TAG=dma2 timeout -k 10 500 python tools/conv_tile_bench.py 5 20 > gpurun_out/r3/d_bench_dma2.log 2>&1; echo "== DMA every 2nd group"; bench gpurun_out/r3/d_bench_dma2.log
for shp in "8 128 128 128 128" "8 64 64 256 256" "8 256 256 64 64" "8 512 512 32 32"; do
  tag=$(echo $shp | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/d_kt_$tag -o p -- python tools/conv_one.py $shp 12 01 > gpurun_out/r3/d_kt_$tag.log 2>&1
  f=$(find gpurun_out/r3/d_kt_$tag -name "*kernel_stats.csv" | head -1); echo "== kernel stats $shp"; grep -i "conv" $f | cut -d, -f1-6 | cut -c1-200
done
shp="8 128 128 128 128"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r3/d_pmc1 -o p -- python tools/conv_one.py $shp 6 01 > gpurun_out/r3/d_pmc1.log 2>&1
python tools/pmc_kernel_table.py $(find gpurun_out/r3/d_pmc1 -name "*counter_collection.csv" | head -1) conv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/r3/d_pmc2 -o p -- python tools/conv_one.py $shp 6 01 > gpurun_out/r3/d_pmc2.log 2>&1
python tools/pmc_kernel_table.py $(find gpurun_out/r3/d_pmc2 -name "*counter_collection.csv" | head -1) conv
rm -rf gpurun_out/r3/d_kt_*/ 2>/dev/null
