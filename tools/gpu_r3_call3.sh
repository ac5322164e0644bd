#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c3; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(MDX_CLOSE_REPORT=1 timeout 1800 python -m pytest tests -m gpu -q -s --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_gpu.log | tail -8
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-cond-scenes 0 --vae-scenes 0 --ops-json $OUT/ops_b128.json > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c3/bench.json'))
print(d['value'], d['ms_per_step'], d['config']['batch_consistency_rel'])
for k,v in d['roofline']['per_kernel'].items(): print(k, v)
PY
