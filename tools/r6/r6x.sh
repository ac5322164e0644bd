run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --no-ops 2>&1 | grep -E "per call"; }
for n in 1 2 4 8; do
run base $n X=1
run res2 $n MDX_ATTN2_RES=2
run d80 $n MDX_ATTN2_D80=1
run qt2 $n MDX_ATTN2_QT=2
done
