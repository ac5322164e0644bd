mkdir -p gpurun_out/r5s
export MDX_PARITY_LOG=$PWD/gpurun_out/r5s/parity_measured.jsonl
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r5s/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5s/pytest_gpu.log)
tail -8 gpurun_out/r5s/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
