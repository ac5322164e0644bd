#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c15; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -k "fused_layernorm" > $OUT/pytest_ln.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_ln.log)
grep -E "passed|failed|rc=" $OUT/pytest_ln.log | tail -3; grep -E "^FAILED|AssertionError" $OUT/pytest_ln.log | head
for rep in 1 2; do
for lib in "" lnab1 lnab3; do
  echo "== lib=${lib:-main} rep $rep" | tee -a $OUT/lnone.log
  if [ -z "$lib" ]; then python tools/lnone.py 2>/dev/null | tee -a $OUT/lnone.log; else MDX_LIB_PATH=$PWD/magicdrive_amd/libmdx_$lib.so python tools/lnone.py 2>/dev/null | tee -a $OUT/lnone.log; fi
done; done
