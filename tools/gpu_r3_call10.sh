#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c10; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_gpu.log | tail -4; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 1500 python bench.py --steps 3 --warmup 1 --ops-json $OUT/ops_b128.json > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c10/bench.json'))
print(d['value'], d['ms_per_step'], d['dtype'], d['config']['batch_consistency_rel'], d['config']['full_cond_scenes_per_s'], d['config']['hires'], d['config']['scenes_per_s_incl_vae_decode'])
print(d['cpu_baseline'])
r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','frac','avg_launch_us')}); print(r['traffic'])
for k,v in r['per_kernel'].items(): print(k, v)
PY
