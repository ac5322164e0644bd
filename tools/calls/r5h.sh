mkdir -p gpurun_out/r5h
timeout 300 python tools/vaeone.py --scenes 1 --json gpurun_out/r5h/vae_ops_1.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5h/vaeone_1.log
timeout 300 python tools/vaeone.py --scenes 4 2>&1 | grep -v amdgpu.ids | head -12 | tee gpurun_out/r5h/vaeone_4.log
