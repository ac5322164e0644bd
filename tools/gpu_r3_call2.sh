#!/bin/bash
# Round-3 GPU call 2: full GPU suite (report mode) on the build with the specialised XL staging, the fixed gemm_ws loop and the
# folded attention; A/B against the previous build (side library), interleaved; a short bench.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c2; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(MDX_CLOSE_REPORT=1 timeout 1500 python -m pytest tests -m gpu -q -s -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log)
tail -6 $OUT/pytest_gpu.log
V=768
PREV=$PWD/magicdrive_amd/libmdx_prev.so
for rep in 1 2; do
  echo "== new rep $rep" >> $OUT/ab.log
  timeout 300 python tools/xlone.py --views $V --reps 5 >> $OUT/ab.log 2>&1
  timeout 300 python tools/kbench.py --views $V --only attn --reps 5 2>&1 | grep -v "T=350\|T=91\|T=28" >> $OUT/ab.log
  echo "== prev rep $rep" >> $OUT/ab.log
  MDX_LIB_PATH=$PREV timeout 300 python tools/xlone.py --views $V --reps 5 >> $OUT/ab.log 2>&1
  MDX_LIB_PATH=$PREV timeout 300 python tools/kbench.py --views $V --only attn --reps 5 2>&1 | grep -v "T=350\|T=91\|T=28" >> $OUT/ab.log
done
grep -v amdgpu.ids $OUT/ab.log
MDX_XL_TIMING=1 timeout 300 python tools/xl_timing.py --views $V 2>&1 | grep -v "quarter\|amdgpu.ids" > $OUT/xl_timing.log
cat $OUT/xl_timing.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-cond-scenes 0 --vae-scenes 0 --ops-json $OUT/ops_b128.json > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
