mkdir -p gpurun_out/r6h
run() { tag=$1; shift; echo "== $tag"; env "$@" python tools/lat1.py --rows-json gpurun_out/r6h/rows_$tag.json 2>&1 | grep -E "per call|op by op"; }
run base X=1
run gn1 MDX_GN_TWO_STAGE=0
run xl2 MDX_GEMM_XL=2
run xlmin32 MDX_XL_MIN_TILES=32
run ws2 MDX_GEMM_WS=2
run bk32 MDX_GEMM_BK=32
run nofork X=1 2>/dev/null
