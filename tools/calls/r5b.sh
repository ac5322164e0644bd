mkdir -p gpurun_out/r5b
{
echo "== main"; timeout 200 python tools/a3_sweep.py
for v in abl1 abl8 abl16 abl24 abl64 abl89 dmatop; do echo "== $v"; MDX_LIB_PATH=$PWD/magicdrive_amd/libmdx_$v.so timeout 200 python tools/a3_sweep.py --attn3 1 --tks 704,1408; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5b/a3_ablate.log
