#!/bin/bash
# Builds debug variants of the library with parts of the convolution kernels removed (LH_ABL bits, see
# igemm_ring_kernel.h) into tools/abl/, for timing experiments only:  LH_LIB_PATH=tools/abl/lib_abl8.so python ...
# usage: tools/ablate.sh 4 8 16 ...   (bf16 kernels only; the other objects are taken from the normal build)
set -e
cd "$(dirname "$0")/../lighthand_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off"
UNITS="${LH_ABL_UNITS:-igemm_ring_bf16_big igemm_ring_bf16_mid igemm_ring_bf16_small wgrad_ring_bf16 igemm_pw_bf16}"
for v in "$@"; do
  for u in $UNITS; do /opt/rocm/bin/hipcc $FLAGS -DLH_ABL=$v -c $u.hip -o /tmp/${u}_abl$v.o & done
  wait
  OTHERS=$(ls *.o | grep -v -E "^($(echo $UNITS | tr ' ' '|'))\.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $(for u in $UNITS; do echo /tmp/${u}_abl$v.o; done) -ldl -o ../../tools/abl/lib_abl$v.so
done
ls -la ../../tools/abl/
