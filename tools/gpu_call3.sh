set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export MDX_PARITY_LOG=$PWD/gpurun_out/r04c_parity_measured.jsonl
rm -f $MDX_PARITY_LOG gpurun_out/r04_xl_bpol_ab.log
for rep in 1 2; do
for lib in libmdx.so libmdx_bpol1.so libmdx_bpol2.so; do
  echo "== $lib (rep $rep)" >> gpurun_out/r04_xl_bpol_ab.log
  MDX_LIB_PATH=$PWD/magicdrive_amd/$lib timeout 300 python tools/xlone.py --views 768 --reps 5 --only c_28x50_640_320,c160_28x50_320_320,c256_14x25_1280_1280,c_14x25_1920_640,c256_7x13,g256_qk_L1,g256_geglu_L1,g256_cc_L1 >> gpurun_out/r04_xl_bpol_ab.log 2>&1
done
done
cat gpurun_out/r04_xl_bpol_ab.log
echo "== ATTN2_D80=1" > gpurun_out/r04_attn_d80.log
MDX_ATTN2_D80=1 timeout 300 python tools/attnone.py --views 768 >> gpurun_out/r04_attn_d80.log 2>&1
cat gpurun_out/r04_attn_d80.log
timeout 900 python -m pytest tests/test_fp16_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "cfg_loop or large_common_offset" -s 2>&1 | tail -12 > gpurun_out/r04c_pytest.log
cat gpurun_out/r04c_pytest.log
timeout 600 python tools/streams_ab.py --pairs 192:3,256:4 > gpurun_out/r04_streams_ab2.log 2>/dev/null
cat gpurun_out/r04_streams_ab2.log
