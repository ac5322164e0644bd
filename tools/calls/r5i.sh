mkdir -p gpurun_out/r5i
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm" --timeout 300 > gpurun_out/r5i/pytest_gn.log 2>&1; echo "rc=$?" >> gpurun_out/r5i/pytest_gn.log)
tail -4 gpurun_out/r5i/pytest_gn.log
(timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "vae" --timeout 600 > gpurun_out/r5i/pytest_vae.log 2>&1; echo "rc=$?" >> gpurun_out/r5i/pytest_vae.log)
tail -4 gpurun_out/r5i/pytest_vae.log
timeout 300 python tools/vaeone.py --scenes 1 2>&1 | grep -v amdgpu.ids | head -16 | tee gpurun_out/r5i/vaeone_1.log
