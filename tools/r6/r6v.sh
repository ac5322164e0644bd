python -m pytest tests/test_kernels_gpu.py -q -x -k "splitk or conv_mfma" 2>&1 | tail -4
run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --no-ops 2>&1 | grep -E "per call"; }
for n in 1 2 4; do
run two $n MDX_SPLITK_FUSED=0
run fused $n X=1
done
