#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c6; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(MDX_CLOSE_REPORT=1 timeout 1200 python -m pytest tests/test_fp16_gpu.py -m gpu -q -s --timeout 900 > $OUT/pytest_fp16.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fp16.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_fp16.log | tail -4; grep -E "^FAILED|^ERROR|Error" $OUT/pytest_fp16.log | head
V=768
SIDE=$PWD/magicdrive_amd/libmdx_r2res.so
for rep in 1 2; do
  echo "== new rep $rep" >> $OUT/ab.log
  timeout 300 python tools/xlone.py --views $V --reps 5 --only c160,g256_ffout,g160_ffout,g256_geglu_L2,g256_cc >> $OUT/ab.log 2>&1
  echo "== side: 320-wide residual in batches of four, rep $rep" >> $OUT/ab.log
  MDX_LIB_PATH=$SIDE timeout 300 python tools/xlone.py --views $V --reps 5 --only c160,g256_ffout,g160_ffout >> $OUT/ab.log 2>&1
  echo "== raster 1 rep $rep" >> $OUT/ab.log
  MDX_XL_RASTER=1 timeout 300 python tools/xlone.py --views $V --reps 5 --only g256_geglu_L2,g256_geglu_L1,c160_28x50_320 >> $OUT/ab.log 2>&1
done
grep -v amdgpu.ids $OUT/ab.log
