mkdir -p gpurun_out/r6k
python -m pytest tests/test_routes_gpu.py tests/test_kernels_gpu.py -q -x 2>&1 | tail -5
run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --rows-json gpurun_out/r6k/rows_${tag}_$n.json 2>&1 | grep -E "per call|op by op"; }
for n in 1 2 4; do
run old $n MDX_GEMM_SMALL_TILES=0
run new $n X=1
run newgn $n MDX_GN_TWO_STAGE=0
done
