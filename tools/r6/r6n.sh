mkdir -p gpurun_out/r6n
python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm" 2>&1 | tail -3
python -m pytest tests/test_e2e_gpu.py -q -x -k "tiny or module_api" 2>&1 | tail -3
for n in 1 2; do echo "== scenes $n"; python tools/lat1.py --scenes $n --rows-json gpurun_out/r6n/rows_$n.json 2>&1 | grep -E "per call|op by op|groupnorm"; done
