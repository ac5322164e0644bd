#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c12; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_gpu.log | tail -4; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
KONE_VIEWS=768 bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc_collect.log 2>&1
cat $OUT/pmc/errors.log 2>/dev/null
python tools/pmc_summarize.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summarize.log 2>&1; tail -3 $OUT/pmc_summarize.log | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +4M -delete
for pz in 1 0; do
MDX_XL_PERSIST=$pz timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-cond-scenes 0 --vae-scenes 0 --hires-scenes 0 --ops-json $OUT/ops_b128_p$pz.json > $OUT/bench_p$pz.json 2> $OUT/bench_p$pz.err; python - <<PY
import json
d=json.load(open('gpurun_out/r3c12/bench_p$pz.json'))
print('XL_PERSIST=$pz', d['value'], d['ms_per_step'], d['config']['batch_consistency_rel'])
for k,v in list(d['roofline']['per_kernel'].items())[:12]: print('   ', k, v['ms_per_step'], v['tflops'])
PY
done
