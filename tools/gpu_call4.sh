set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MDX_PARITY_LOG=$PWD/gpurun_out/r04d_parity_measured.jsonl
rm -f $MDX_PARITY_LOG
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_routes_gpu.py -m gpu -q -k "attn or attention" 2>&1 | tail -15 > gpurun_out/r04d_pytest_attn.log
tail -4 gpurun_out/r04d_pytest_attn.log
for res in 0 1; do
  echo "== ATTN2_RES=$res" >> gpurun_out/r04_attn_res_ab.log
  MDX_ATTN2_RES=$res timeout 300 python tools/attnone.py --views 768 >> gpurun_out/r04_attn_res_ab.log 2>&1
done
cat gpurun_out/r04_attn_res_ab.log
timeout 900 python tools/streams_ab.py --pairs 192:1,192:2,192:3,192:4 > gpurun_out/r04_streams_ab3.log 2>/dev/null
cat gpurun_out/r04_streams_ab3.log
