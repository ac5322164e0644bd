REPO=$PWD; mkdir -p gpurun_out/r6u
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/r6u/stats -o lat1 --output-format csv -- python $REPO/tools/lat1.py --no-ops > $REPO/gpurun_out/r6u/lat1.log 2>&1
cd $REPO
grep "per call" gpurun_out/r6u/lat1.log
f=$(find gpurun_out/r6u/stats -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-160
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r6u/stats/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 50-step call: take the last 568*50 kernels
n = 568 * 50
last = rows[-n:]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last)
wall = int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])
# union of busy intervals (two streams overlap)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in last)
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print(f"last call: {len(last)} kernels, wall {wall/1e6:.2f} ms, sum of kernel durations {busy/1e6:.2f} ms, union of busy intervals {u/1e6:.2f} ms, idle {100*(wall-u)/wall:.1f} %")
PY
find gpurun_out/r6u -name "*kernel_trace.csv" -delete
