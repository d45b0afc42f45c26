set -o pipefail
export TMPDIR=/tmp; mkdir -p gpurun_out/v1
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parallel.py -x -q > gpurun_out/v1/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/v1/tests.log
bash tools/gpu_ab.sh "EGM_DEFER_BGRAD=0" "EGM_DEFER_BGRAD=1" 2>&1 | tee gpurun_out/v1/ab_bgrad.txt
timeout -k 10 300 python tools/conv_tile_bench.py 5 20 > gpurun_out/v1/tile_product.txt 2>&1; echo "tile product rc=$?"
EGM_LIB_TAG=mfma16 timeout -k 10 300 python tools/conv_tile_bench.py 5 20 > gpurun_out/v1/tile_mfma16.txt 2>&1; echo "tile mfma16 rc=$?"
bash tools/gpu_ab.sh "EGM_LIB_TAG=" "EGM_LIB_TAG=mfma16" 2>&1 | tee gpurun_out/v1/ab_mfma16.txt
EGM_LIB_TAG=timing timeout -k 10 200 python tools/conv_tile_diag.py 8 512 512 32 64 > gpurun_out/v1/diag_32_64_512.txt 2>&1; echo "diag rc=$?"
EGM_LIB_TAG=timing timeout -k 10 200 python tools/conv_tile_diag.py 8 256 256 64 64 > gpurun_out/v1/diag_64_64_256.txt 2>&1
EGM_LIB_TAG=timing timeout -k 10 200 python tools/conv_tile_diag.py 8 128 128 128 128 > gpurun_out/v1/diag_128_128_128.txt 2>&1
tail -5 gpurun_out/v1/tile_mfma16.txt
