#!/bin/bash
# Round-3 GPU call 1: parity of the cleaned-up library (new reference goldens; report mode for the tolerance table) + A/B microbenchmarks.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c1; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(MDX_CLOSE_REPORT=1 timeout 1500 python -m pytest tests/test_sd15_golden_gpu.py tests/test_kernels_gpu.py tests/test_routes_gpu.py tests/test_integration_gpu.py -m gpu -q -s -x --timeout 900 > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log)
tail -5 $OUT/pytest_subset.log
V=768
echo "== xlone raster 2 (default)" > $OUT/xlone.log
timeout 300 python tools/xlone.py --views $V --reps 5 >> $OUT/xlone.log 2>&1
echo "== xlone raster 1 (round-2 order)" >> $OUT/xlone.log
MDX_XL_RASTER=1 timeout 300 python tools/xlone.py --views $V --reps 5 --only c160_28x50_320,c256_14,g256 >> $OUT/xlone.log 2>&1
echo "== ablate: no epilogue (dbg 4)" >> $OUT/xlone.log
MDX_LIB_PATH=$PWD/magicdrive_amd/libmdx_xl_ablate.so MDX_XL_DBG=4 timeout 300 python tools/xlone.py --views $V --reps 5 --only c160_28x50_320,c256_14,g256 >> $OUT/xlone.log 2>&1
echo "== ablate: staging but no stores (dbg 8)" >> $OUT/xlone.log
MDX_LIB_PATH=$PWD/magicdrive_amd/libmdx_xl_ablate.so MDX_XL_DBG=8 timeout 300 python tools/xlone.py --views $V --reps 5 --only c160_28x50_320,c256_14,g256 >> $OUT/xlone.log 2>&1
echo "== gelu: polynomial (default build)" >> $OUT/xlone.log
timeout 200 python tools/xlone.py --views $V --reps 5 --only g_geglu_L0,g256_geglu >> $OUT/xlone.log 2>&1
echo "== gelu: Abramowitz-Stegun side build" >> $OUT/xlone.log
MDX_LIB_PATH=$PWD/magicdrive_amd/libmdx_gelu_as.so timeout 200 python tools/xlone.py --views $V --reps 5 --only g_geglu_L0,g256_geglu >> $OUT/xlone.log 2>&1
MDX_XL_TIMING=1 timeout 300 python tools/xl_timing.py --views $V > $OUT/xl_timing.log 2>&1
cat $OUT/xlone.log
tail -40 $OUT/xl_timing.log
