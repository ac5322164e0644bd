mkdir -p gpurun_out/r5y
timeout 170 python -m pytest tests/test_e2e_gpu.py tests/test_routes_gpu.py -q -k "batch_size_sweep or xl320_edge" --timeout 160 > gpurun_out/r5y/pytest.log 2>&1; tail -5 gpurun_out/r5y/pytest.log | cut -c1-400
