mkdir -p gpurun_out/r5t
timeout 125 python tools/streams_ab.py --ddim-steps 20 --pairs 32:2,32:1:F,32:1,48:2,48:1:F,64:2,64:1:F,96:2,96:1:F > gpurun_out/r5t/mid_batch_policy.log 2> gpurun_out/r5t/err.log; echo rc=$?; cat gpurun_out/r5t/mid_batch_policy.log | cut -c1-230; tail -2 gpurun_out/r5t/err.log | cut -c1-200
