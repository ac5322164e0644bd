mkdir -p gpurun_out/r5p
(timeout 1200 python -m pytest tests/test_e2e_gpu.py -q -x -m gpu --timeout 900 -k "loop_tiny or goldens or unipc or given_view or batch_consistency or sample_driver_end" > gpurun_out/r5p/pytest_e2e.log 2>&1; echo "rc=$?" >> gpurun_out/r5p/pytest_e2e.log)
tail -4 gpurun_out/r5p/pytest_e2e.log
{ for n in 1 2 4 8 16; do for f in 0 64; do timeout 300 python tools/lat1.py --scenes $n --fork-max $f --no-ops 2>&1 | grep "per call"; done; done; } | tee gpurun_out/r5p/fork_ab.log
