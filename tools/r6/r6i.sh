mkdir -p gpurun_out/r6i
python -m pytest tests/test_routes_gpu.py -q -x -k "three_stages or stage_choice" 2>&1 | tail -15
run() { tag=$1; shift; echo "== $tag"; env "$@" python tools/lat1.py --rows-json gpurun_out/r6i/rows_$tag.json 2>&1 | grep -E "per call|op by op"; }
run s1 MDX_GEMM_STAGES=1
run s0 MDX_GEMM_STAGES=0
run s0gn MDX_GEMM_STAGES=0 MDX_GN_TWO_STAGE=0
