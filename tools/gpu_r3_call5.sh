#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c5; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(timeout 2400 python -m pytest tests -m gpu -q -s --timeout 900 --deselect tests/test_fp16_gpu.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_gpu.log | tail -4; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
(MDX_CLOSE_REPORT=1 timeout 1200 python -m pytest tests/test_fp16_gpu.py -m gpu -q -s --timeout 900 > $OUT/pytest_fp16.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_fp16.log)
grep -E "passed|failed|error|rc=|fp16" $OUT/pytest_fp16.log | tail -12
for dt in bf16 fp16; do
timeout 900 python bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --full-cond-scenes 0 --vae-scenes 0 --ops-json $OUT/ops_b128_$dt.json > $OUT/bench_$dt.json 2> $OUT/bench_$dt.err; python - <<PY
import json
d=json.load(open('gpurun_out/r3c5/bench_$dt.json'))
print('$dt', d['value'], d['ms_per_step'], d['config']['batch_consistency_rel'], d['dtype'])
for k,v in list(d['roofline']['per_kernel'].items())[:10]: print('   ', k, v['ms_per_step'], v['tflops'])
PY
done
