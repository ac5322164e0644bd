mkdir -p gpurun_out/r5e
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention3 or attention2" --timeout 300 > gpurun_out/r5e/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r5e/pytest_attn.log)
tail -5 gpurun_out/r5e/pytest_attn.log
{ timeout 200 python tools/a3_sweep.py; timeout 200 python tools/a3_sweep.py --xview --tks 1400; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e/a3_sweep.log
