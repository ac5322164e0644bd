mkdir -p gpurun_out/r06z2
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r06z2/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06z2/pytest_gpu.log)
grep -E "passed|failed|rc=" gpurun_out/r06z2/pytest_gpu.log | tail -3
timeout 1500 python bench.py --ops-json gpurun_out/r06z2/ops_b192.json > gpurun_out/r06z2/bench_default.json 2> gpurun_out/r06z2/bench.err; tail -c 300 gpurun_out/r06z2/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06z2/bench_default.json'))
r=d['roofline']; c=d['config']
print(d['value'], d['steps'], d['warmup'], c['latency_1scene_s'], c.get('latency_1scene_full_cond_cfg_s'), r['frac'], r['traffic'] if not isinstance(r['traffic'],dict) else {k:r['traffic'][k] for k in list(r['traffic'])[:8]}, r['pmc_status'], r['mfma_util_time_weighted'])
PY
