run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --no-ops 2>&1 | grep -E "per call"; }
for n in 1 2; do
run xl160 $n X=1
run xl64 $n MDX_XL_MIN_TILES=64
run xl32 $n MDX_XL_MIN_TILES=32
done
run xl32 4 MDX_XL_MIN_TILES=32
run xl32 8 MDX_XL_MIN_TILES=32
