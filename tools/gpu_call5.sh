set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MDX_PARITY_LOG=$PWD/gpurun_out/r04e_parity_measured.jsonl
rm -f $MDX_PARITY_LOG
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_routes_gpu.py -m gpu -q -k "attention2 or conv_out or conv_direct or flattened" 2>&1 | tail -15 > gpurun_out/r04e_pytest.log
tail -4 gpurun_out/r04e_pytest.log
for vm in 0 1; do for res in 0 1; do
  echo "== ATTN2_VIEWMAP=$vm ATTN2_RES=$res" >> gpurun_out/r04_attn_viewmap_ab.log
  MDX_ATTN2_VIEWMAP=$vm MDX_ATTN2_RES=$res timeout 300 python tools/attnone.py --views 768 >> gpurun_out/r04_attn_viewmap_ab.log 2>&1
done; done
grep -v amdgpu gpurun_out/r04_attn_viewmap_ab.log
echo "== new col_split epilogue / conv_out WS" > gpurun_out/r04_vt_convout_ab.log
timeout 300 python tools/xlone.py --views 768 --reps 5 --only vt_,convout >> gpurun_out/r04_vt_convout_ab.log 2>&1
echo "== CONV_OUT_WS=0, old attention lib (old col_split epilogue)" >> gpurun_out/r04_vt_convout_ab.log
MDX_CONV_OUT_WS=0 MDX_LIB_PATH=$PWD/magicdrive_amd/libmdx_a2old.so timeout 300 python tools/xlone.py --views 768 --reps 5 --only vt_,convout >> gpurun_out/r04_vt_convout_ab.log 2>&1
grep -v amdgpu gpurun_out/r04_vt_convout_ab.log
