mkdir -p gpurun_out/r5c
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" --timeout 300 > gpurun_out/r5c/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r5c/pytest_attn.log)
tail -15 gpurun_out/r5c/pytest_attn.log
{ timeout 200 python tools/a3_sweep.py; timeout 200 python tools/a3_sweep.py --xview --tks 1400; MDX_ATTN3_WALK=1 timeout 200 python tools/a3_sweep.py --attn3 1 --tks 704,1408; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c/a3_sweep.log
