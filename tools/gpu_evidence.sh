#!/bin/bash
# One gpurun call's worth of round evidence (≈ 25 GPU-minutes): GPU suite, the bench command (+ per-op table), kernel statistics of the bench command (eager
# launches), counter passes (tools/pmc_collect.sh) summarised with the library's build id.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/evidence}; mkdir -p $OUT
REPO=$PWD
export MDX_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl
rm -f $MDX_PARITY_LOG
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log)
grep -E "passed|failed|error|rc=" $OUT/pytest_gpu.log | tail -4; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
unset MDX_PARITY_LOG
timeout 1500 python bench.py --steps ${BENCH_STEPS:-6} --warmup ${BENCH_WARMUP:-2} --ops-json $OUT/ops_b192.json > $OUT/bench_b192.json 2> $OUT/bench_b192.err; tail -c 600 $OUT/bench_b192.err; OUT=$OUT python - <<'PY'
import json, os
d=json.load(open(os.environ['OUT']+'/bench_b192.json'))
print({k:v for k,v in d.items() if k not in ('roofline','config','cpu_baseline')}); print(d['cpu_baseline']); print({k:v for k,v in d['config'].items() if k!='workload'})
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('per_kernel','per_family','traffic')})
for k,v in r['per_kernel'].items(): print('   %-48s %8.3f ms %5d launches %8.1f us %s TF  mfma_util=%s hbm=%s'%(k, v['ms_per_step'], v['launches'], v['avg_launch_us'], v['tflops'], v.get('mfma_util'), v.get('hbm_gbps')))
PY
# kernel statistics of the bench command with eager launches on ONE stream at the per-stream batch (96 scenes: a launch's average there is its
# isolated time, comparable with the HIP-event figure of the line); STATS_2STREAMS=1 adds the default two-stream run (kernels of the two chunks share the chip)
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -o bench --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-graph --ddim-steps 10 --streams 1 --scenes-per-gpu 96 --no-cpu-baseline --no-op-profile --no-consistency-check --full-cond-scenes 0 --vae-scenes 0 --hires-scenes 0 > $REPO/$OUT/bench_nograph.log 2>&1 )
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
if [ "${STATS_2STREAMS:-0}" = "1" ]; then
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats2 -o bench --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-graph --ddim-steps 10 --no-cpu-baseline --no-op-profile --no-consistency-check --full-cond-scenes 0 --vae-scenes 0 --hires-scenes 0 > $REPO/$OUT/bench_nograph2.log 2>&1 )
fi
find $OUT -name "*kernel_trace.csv" -size +4M -delete
# counters: one --pmc group per pass over tools/kall.py (the bench's per-stream batch), summary tied to the library's build id
bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc_collect.log 2>&1
python tools/pmc_summarize.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summarize.log 2>&1; tail -3 $OUT/pmc_summarize.log
find $OUT/pmc -name "*kernel_trace.csv" -size +1M -delete; find $OUT/pmc -name "*.csv" -size +64k -exec gzip -f {} \;
du -sh $OUT
