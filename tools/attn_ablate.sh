#!/bin/bash
# Build side copies of libmdx.so with attention2.hip ablated (-DA2_ABL=bits) and time them with tools/attnone.py (run on the GPU box
# after `bash tools/attn_ablate.sh build` here).  Results of ablated kernels are wrong by design; this only attributes time.
set -e
cd "$(dirname "$0")/.."
C=magicdrive_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form"
BITS="${BITS:-1 2 4 8 16 32 3 7 24 63}"
if [ "$1" = build ]; then
  mkdir -p tools/ubench/abl
  for b in $BITS; do
    /opt/rocm/bin/hipcc $FLAGS -DA2_ABL=$b -c $C/attention2.hip -o tools/ubench/abl/attention2_$b.o
    objs=$(ls $C/*.o | grep -v attention2.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/abl/libmdx_$b.so $objs tools/ubench/abl/attention2_$b.o
    rm tools/ubench/abl/attention2_$b.o
  done
else
  echo "== product"; python tools/attnone.py --reps 3 | head -2
  for b in $BITS; do echo "== A2_ABL=$b"; MDX_LIB_PATH=tools/ubench/abl/libmdx_$b.so python tools/attnone.py --reps 3 | head -2; done
fi
