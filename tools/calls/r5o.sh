mkdir -p gpurun_out/r5o
REPO=$PWD; OUT=gpurun_out/r5o
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -o bench --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-graph --ddim-steps 10 --streams 1 --scenes-per-gpu 96 --no-cpu-baseline --no-op-profile --no-consistency-check --full-cond-scenes 0 --vae-scenes 0 --hires-scenes 0 > $REPO/$OUT/bench_nograph.log 2>&1 )
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160
find $OUT -name "*kernel_trace.csv" -size +4M -delete
