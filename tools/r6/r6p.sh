run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --no-ops 2>&1 | grep -E "per call"; }
for n in 8 16; do
run xl160 $n X=1
run xl96 $n MDX_XL_MIN_TILES=96
run xl48 $n MDX_XL_MIN_TILES=48
done
run nofork 1 X=1
env python tools/lat1.py --scenes 1 --no-ops --fork-max 0 2>&1 | grep "per call"
