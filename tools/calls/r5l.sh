mkdir -p gpurun_out/r5l
export MDX_PARITY_LOG=$PWD/gpurun_out/r5l/parity_measured.jsonl
(timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_fp16_gpu.py tests/test_integration_gpu.py tests/test_kernels_gpu.py tests/test_routes_gpu.py tests/test_sd15_golden_gpu.py -m gpu -q --timeout 900 -k "not (forward_tiny or loop_tiny or one_pass or reference_golden or module_api or vae_ or hip_vae or real_size or batch_consistency)" > gpurun_out/r5l/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5l/pytest_gpu.log)
tail -12 gpurun_out/r5l/pytest_gpu.log
