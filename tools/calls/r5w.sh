OUT=gpurun_out/r5w; mkdir -p $OUT
bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc_collect.log 2>&1
python tools/pmc_summarize.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summarize.log 2>&1; tail -3 $OUT/pmc_summarize.log
find $OUT/pmc -name "*kernel_trace.csv" -size +1M -delete; find $OUT/pmc -name "*.csv" -size +64k -exec gzip -f {} \;
[ -s $OUT/pmc_summary.json ] && cp $OUT/pmc_summary.json profiles/r05_pmc_summary.json
timeout 420 python bench.py --steps 4 --warmup 1 --ops-json $OUT/ops_b192.json > $OUT/bench_b192.json 2> $OUT/bench_b192.err; tail -c 300 $OUT/bench_b192.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5w/bench_b192.json'))
print({k:v for k,v in d.items() if k not in ('roofline','config','cpu_baseline')})
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('per_kernel','per_family')})
PY
