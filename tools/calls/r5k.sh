mkdir -p gpurun_out/r5k
export MDX_PARITY_LOG=$PWD/gpurun_out/r5k/parity_measured.jsonl
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > gpurun_out/r5k/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5k/pytest_gpu.log)
tail -8 gpurun_out/r5k/pytest_gpu.log
