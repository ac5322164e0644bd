cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "libmdx.so 0" "libmdx.so 1" "libmdx_lb4two.so 1" "libmdx.so 0" "libmdx_lb4two.so 1"; do
  set -- $cfg
  echo "== $1 ATTN2_QT=$2" >> gpurun_out/r04_attn_xview_lb4_ab.log
  MDX_LIB_PATH=$PWD/magicdrive_amd/$1 MDX_ATTN2_QT=$2 timeout 300 python tools/attnone.py --views 576 2>&1 | grep "d=40" >> gpurun_out/r04_attn_xview_lb4_ab.log
done
cat gpurun_out/r04_attn_xview_lb4_ab.log
