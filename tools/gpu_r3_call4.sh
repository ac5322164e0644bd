#!/bin/bash
# Round-3 GPU call 4: counters at the bench batch (768 views), kernel statistics of the bench command, attention A/B.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3c4; mkdir -p $OUT
REPO=$PWD
for cfg in "ATTN2_QT=2" "ATTN2_QT=1" "ATTN2_FOLD=0"; do
  echo "== $cfg" >> $OUT/attn_ab.log
  env MDX_$cfg timeout 300 python tools/kbench.py --views 768 --only attn --reps 5 --prescaled 2>&1 | grep "T=1400" >> $OUT/attn_ab.log
done
cat $OUT/attn_ab.log
KONE_VIEWS=768 bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc_collect.log 2>&1
cat $OUT/pmc/errors.log 2>/dev/null
python tools/pmc_summarize.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summarize.log 2>&1; tail -25 $OUT/pmc_summarize.log | cut -c1-400
# kernel statistics of the bench command at the bench batch (eager launches: rocprofv3 segfaults under hipGraph replay on this image)
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -o bench --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-graph --ddim-steps 10 --no-cpu-baseline --no-op-profile --no-consistency-check --full-cond-scenes 0 --vae-scenes 0 > $REPO/$OUT/bench_nograph.log 2>&1 )
ls $OUT/stats 2>/dev/null | head; f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-220
# keep the merged output small: drop the raw traces
find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete; du -sh $OUT
