mkdir -p gpurun_out/r5q
timeout 900 python tools/streams_ab.py --pairs 192:2,192:2:f,192:2,192:2:f 2>&1 | grep "^{" | tee gpurun_out/r5q/fork_chunks_ab.log
