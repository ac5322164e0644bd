echo "--- stages 1"; MDX_GEMM_STAGES=1 python tools/r6/ktiming_small.py 2>&1 | grep gemm_conv
echo "--- stages 3"; MDX_GEMM_STAGES=3 python tools/r6/ktiming_small.py 2>&1 | grep gemm_conv
run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --no-ops 2>&1 | grep -E "per call"; }
run s1 1 MDX_GEMM_STAGES=1
run s3 1 MDX_GEMM_STAGES=3
