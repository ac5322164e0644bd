set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04g; mkdir -p $OUT
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_gpu.log | tail -4; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
unset MDX_PARITY_LOG
timeout 1200 python bench.py --steps 3 --warmup 1 --ops-json $OUT/ops_b192.json > $OUT/bench_b192.json 2> $OUT/bench_b192.err; tail -c 400 $OUT/bench_b192.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g/bench_b192.json'))
print({k:v for k,v in d.items() if k not in ('roofline','config','cpu_baseline')}); print({k:v for k,v in d['config'].items() if k!='workload'})
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('per_kernel','per_family','traffic')})
for k,v in r['per_kernel'].items(): print('   %-48s %8.3f ms %5d launches %8.1f us %s TF'%(k, v['ms_per_step'], v['launches'], v['avg_launch_us'], v['tflops']))
print(r['per_family'])
PY
