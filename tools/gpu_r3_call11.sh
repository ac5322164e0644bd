#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c11; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(timeout 1500 python -m pytest tests/test_routes_gpu.py tests/test_fp16_gpu.py -m gpu -q --timeout 600 -x -k "xl_gemm or persistent or fp16_gemm or forced_xl or geglu" > $OUT/pytest_xlp.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_xlp.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_xlp.log | tail -4; grep -E "^FAILED|^ERROR|Error|assert" $OUT/pytest_xlp.log | head -20
V=768
for rep in 1 2; do
  for pz in 1 0; do
    echo "== XL_PERSIST=$pz rep $rep" >> $OUT/ab.log
    MDX_XL_PERSIST=$pz timeout 300 python tools/xlone.py --views $V --reps 5 --only g256 >> $OUT/ab.log 2>&1
  done
done
grep -v amdgpu.ids $OUT/ab.log
