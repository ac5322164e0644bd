mkdir -p gpurun_out/r6o
python -m pytest tests/test_kernels_gpu.py tests/test_routes_gpu.py -q -x 2>&1 | tail -3
for n in 1 2 4; do echo "== scenes $n"; python tools/lat1.py --scenes $n --rows-json gpurun_out/r6o/rows_$n.json 2>&1 | grep -E "per call|op by op"; done
