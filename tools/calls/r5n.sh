mkdir -p gpurun_out/r5n
timeout 600 python tools/lat1.py --scenes 1 2>&1 | grep -v "amdgpu.ids\|Warning" | tee gpurun_out/r5n/lat1.log
timeout 600 python tools/lat1.py --scenes 4 2>&1 | grep -v "amdgpu.ids\|Warning" | head -12 | tee gpurun_out/r5n/lat4.log
