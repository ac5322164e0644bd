mkdir -p gpurun_out/r5g
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention2" --timeout 300 > gpurun_out/r5g/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/pytest_attn.log)
tail -3 gpurun_out/r5g/pytest_attn.log
(MDX_ATTN2_PERSIST=4 MDX_ATTN3=0 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention2 or joint" --timeout 300 > gpurun_out/r5g/pytest_attn_p4.log 2>&1; echo "rc=$?" >> gpurun_out/r5g/pytest_attn_p4.log)
tail -3 gpurun_out/r5g/pytest_attn_p4.log
{
for pz in 0 2 3 4 5 6; do echo "== ATTN2_PERSIST=$pz"; MDX_ATTN2_PERSIST=$pz MDX_ATTN3=0 timeout 300 python tools/attnone.py --views 576 2>&1 | tail -5; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5g/attn2_persist.log
