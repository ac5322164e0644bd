set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MDX_PARITY_LOG=$PWD/gpurun_out/r04f_parity_measured.jsonl
rm -f $MDX_PARITY_LOG
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_routes_gpu.py -m gpu -q -k "attn or attention or conv_out or conv_direct" 2>&1 | tail -15 > gpurun_out/r04f_pytest.log
tail -4 gpurun_out/r04f_pytest.log
for sw in 1 2; do
  echo "== ATTN_SWZ=$sw" >> gpurun_out/r04_attn_swz_ab.log
  MDX_ATTN_SWZ=$sw timeout 300 python tools/attnone.py --views 768 >> gpurun_out/r04_attn_swz_ab.log 2>&1
done
grep -v amdgpu gpurun_out/r04_attn_swz_ab.log
timeout 300 python tools/xlone.py --views 768 --reps 5 --only convout 2>&1 | grep -v amdgpu
