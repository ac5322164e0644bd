set -u
mkdir -p gpurun_out/r5a
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" --timeout 300 > gpurun_out/r5a/pytest_attn.log 2>&1; echo "rc=$?" >> gpurun_out/r5a/pytest_attn.log)
tail -15 gpurun_out/r5a/pytest_attn.log
for a3 in 0 1; do echo "== ATTN3=$a3"; MDX_ATTN3=$a3 timeout 300 python tools/attnone.py --views 576 2>&1 | tail -6; done | tee gpurun_out/r5a/attnone_ab.log
