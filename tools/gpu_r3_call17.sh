#!/bin/bash
# Round-3 last call (3 GPU-minutes left): LayerNorm kernel after the compile-time AFFINE switch — tests, then one bench call for the per-op times.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c17; mkdir -p $OUT
(timeout 70 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 60 -k "test_layernorm or fused_layernorm" > $OUT/pytest_ln.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_ln.log)
grep -E "passed|failed|rc=" $OUT/pytest_ln.log | tail -2
timeout 95 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-consistency-check --full-cond-scenes 0 --vae-scenes 0 --hires-scenes 0 --ops-json $OUT/ops_b128.json > $OUT/bench_b128.json 2> $OUT/bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r3c17/bench_b128.json')); print(d['value'], d['ms_per_step'], d['config']['mfma_frac_end_to_end'])
    o=json.load(open('gpurun_out/r3c17/ops_b128.json')); ln=[r for r in o if r['kernel']=='layernorm_kernel']; print(len(ln), sum(r['ms'] for r in ln))
except Exception as e: print('no bench', e)
PY
