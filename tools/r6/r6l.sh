mkdir -p gpurun_out/r6l
python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm" 2>&1 | tail -3
run() { tag=$1; n=$2; shift; shift; echo "== $tag scenes $n"; env "$@" python tools/lat1.py --scenes $n --rows-json gpurun_out/r6l/rows_${tag}_$n.json 2>&1 | grep -E "per call|op by op"; }
for n in 1 2 3 5 7; do
run c3on $n X=1
run c3off $n MDX_CONV3=0
done
