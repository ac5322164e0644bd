set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MDX_PARITY_LOG=$PWD/gpurun_out/r04a_parity_measured.jsonl
rm -f $MDX_PARITY_LOG
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_sd15_golden_gpu.py tests/test_fp16_gpu.py -m gpu -q -x -k "given_view or cxyz or cfg_loop or unipc" -s 2>&1 | tail -40 > gpurun_out/r04a_pytest_new.log
tail -5 gpurun_out/r04a_pytest_new.log
timeout 900 python tools/streams_ab.py > gpurun_out/r04_streams_ab.log 2>gpurun_out/r04_streams_ab.err
cat gpurun_out/r04_streams_ab.log; tail -3 gpurun_out/r04_streams_ab.err
