#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c14; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_sd15_golden_gpu.py tests/test_integration_gpu.py -m gpu -q --timeout 600 -x \
   -k "layernorm or weight_stationary or fused_qkv or fp16_gemm_conv_norm or sd15 or fp16_sd15 or loop or forward" > $OUT/pytest_ln.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_ln.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_ln.log | tail -4; grep -E "^FAILED|^ERROR|Error|assert" $OUT/pytest_ln.log | head -20
for lf in 1 0; do
MDX_LN_FUSE=$lf timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-cond-scenes 0 --vae-scenes 0 --hires-scenes 0 --ops-json $OUT/ops_b128_lf$lf.json > $OUT/bench_lf$lf.json 2> $OUT/bench_lf$lf.err; python - <<PY
import json
d=json.load(open('$OUT/bench_lf$lf.json'))
print('LN_FUSE=$lf', d['value'], d['ms_per_step'], d['config']['batch_consistency_rel'])
ops=json.load(open('$OUT/ops_b128_lf$lf.json'))
PY
done
python - <<'PY'
import json, collections
for lf in (1,0):
    o=json.load(open(f'gpurun_out/r3c14/ops_b128_lf{lf}.json'))
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in o:
        if 'ws' in r['kernel'] or 'layernorm' in r['kernel']:
            agg[r['kernel']][0]+=r['ms']; agg[r['kernel']][1]+=1
    print('LN_FUSE',lf, {k:(round(v[0],3),v[1]) for k,v in agg.items()}, 'sum', round(sum(v[0] for v in agg.values()),3), 'total', round(sum(r['ms'] for r in o),2))
    if lf==1:
        for r in o:
            if "ln" in r["kernel"] and "unet.d0.a0.tb0" in r["name"]: print('   ', r['name'], r['kernel'], round(r['ms'],4))
PY
