mkdir -p gpurun_out/r5m
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_routes_gpu.py -q -x -k "geglu" --timeout 300 > gpurun_out/r5m/pytest_geglu.log 2>&1; echo "rc=$?" >> gpurun_out/r5m/pytest_geglu.log)
tail -3 gpurun_out/r5m/pytest_geglu.log
{ for lib in libmdx_nopk.so libmdx.so libmdx_nopk.so libmdx.so; do echo "== $lib"; MDX_LIB_PATH=$PWD/magicdrive_amd/$lib timeout 300 python tools/xlone.py --views 576 --only g256_geglu_L1,g256_geglu_L2,g_geglu_L0 --reps 10; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5m/gelu_pk_ab.log
