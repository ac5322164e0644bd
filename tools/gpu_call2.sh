set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MDX_PARITY_LOG=$PWD/gpurun_out/r04b_parity_measured.jsonl
rm -f $MDX_PARITY_LOG
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_routes_gpu.py -m gpu -q -k "attn" 2>&1 | tail -15 > gpurun_out/r04b_pytest_attn.log
tail -3 gpurun_out/r04b_pytest_attn.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_sd15_golden_gpu.py tests/test_fp16_gpu.py -m gpu -q -k "not kernels" -s 2>&1 | tail -60 > gpurun_out/r04b_pytest_e2e.log
tail -8 gpurun_out/r04b_pytest_e2e.log
for lib in libmdx_a2old.so libmdx.so; do
  echo "== $lib" >> gpurun_out/r04_attn_issue_ab.log
  MDX_LIB_PATH=$PWD/magicdrive_amd/$lib timeout 300 python tools/attnone.py --views 768 >> gpurun_out/r04_attn_issue_ab.log 2>&1
done
for qt in 1 2; do
  echo "== libmdx.so ATTN2_QT=$qt" >> gpurun_out/r04_attn_issue_ab.log
  MDX_ATTN2_QT=$qt timeout 300 python tools/attnone.py --views 768 >> gpurun_out/r04_attn_issue_ab.log 2>&1
done
cat gpurun_out/r04_attn_issue_ab.log
