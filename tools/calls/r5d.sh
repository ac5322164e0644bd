mkdir -p gpurun_out/r5d
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" --timeout 300 > gpurun_out/r5d/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r5d/pytest_attn.log)
tail -25 gpurun_out/r5d/pytest_attn.log
{ timeout 200 python tools/a3_sweep.py; timeout 200 python tools/a3_sweep.py --xview --tks 1400; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5d/a3_sweep.log
