mkdir -p gpurun_out/r5t
timeout 900 python tools/streams_ab.py --pairs 24:1,24:1:F,32:1,32:1:F,32:2,48:1:F,48:2,64:1:F,64:2,96:1:F,96:2 > gpurun_out/r5t/mid_batch_policy.log 2> gpurun_out/r5t/err.log
grep "^{" gpurun_out/r5t/mid_batch_policy.log; tail -5 gpurun_out/r5t/err.log
