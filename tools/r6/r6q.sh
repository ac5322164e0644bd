run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --no-ops 2>&1 | grep -E "per call"; }
for n in 3 4 6 12; do
run xl160 $n X=1
run xl96 $n MDX_XL_MIN_TILES=96
run xl64 $n MDX_XL_MIN_TILES=64
done
run xl128 8 MDX_XL_MIN_TILES=128
run xl64 8 MDX_XL_MIN_TILES=64
