mkdir -p gpurun_out/r5j
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" --timeout 300 > gpurun_out/r5j/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r5j/pytest_attn.log)
tail -6 gpurun_out/r5j/pytest_attn.log
{ for pf in 0 1 0 1; do echo "== ATTN2_PF=$pf"; MDX_ATTN2_PF=$pf timeout 300 python tools/attnone.py --views 576 2>&1 | grep "d=40"; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5j/attn2_pf_ab.log
