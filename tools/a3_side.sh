#!/bin/bash
# side libraries that differ from the main build only in attention3.hip: tools/a3_side.sh NAME "-DA3_ABL=8 ..."  ->  magicdrive_amd/libmdx_NAME.so
set -e
cd "$(dirname "$0")/../magicdrive_amd/csrc"
NAME=$1; FLAGS=$2
mkdir -p build_$NAME
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function"
$CXX $FLAGS -DMDX_ATTN3 -I. -c ../../tools/attn3/attention3.hip -o build_$NAME/attention3.o &
$CXX $FLAGS -DMDX_F16=1 -Dmdx=mdx_f16 -DMDX_ATTN3 -I. -c ../../tools/attn3/attention3.hip -o build_$NAME/attention3_f16.o &
wait
OBJS=$(ls *.o | grep -v '^attention3' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmdx_$NAME.so $OBJS build_$NAME/attention3.o build_$NAME/attention3_f16.o
echo built ../libmdx_$NAME.so
