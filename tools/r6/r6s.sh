run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --no-ops 2>&1 | grep -E "per call"; }
for n in 1 2; do
run ws1 $n X=1
run ws0 $n MDX_GEMM_WS=0
run ws0_nolnfuse $n MDX_GEMM_WS=0 MDX_LN_FUSE=0
done
