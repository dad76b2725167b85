#!/bin/bash
# Debug builds of the convolution + BatchNorm + ReLU launch with parts of its in-launch finalize removed (LH_BNF_ABL bits,
# igemm_epilogue.h) into tools/abl/, for timing only:  LH_LIB_PATH=tools/abl/lib_bnf3.so LH_FUSE_BN_TRAIN=1 python tools/layer_profile.py
# usage: tools/ablate_bnfuse.sh 1 3 7 15 ...   (only the dense-wave bf16 instantiations are rebuilt: what R50's stages 3-4 run)
set -e
cd "$(dirname "$0")/../lighthand_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off"
for v in "$@"; do
  for f in igemm_ring_bf16_dense igemm_ring_bf16_mid; do
    /opt/rocm/bin/hipcc $FLAGS -DLH_BNF_ABL=$v -c $f.hip -o /tmp/${f}_bnf$v.o &
  done
  wait
  OTHERS=$(ls *.o | grep -v "^igemm_ring_bf16_dense.o$\|^igemm_ring_bf16_mid.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/igemm_ring_bf16_dense_bnf$v.o /tmp/igemm_ring_bf16_mid_bnf$v.o -ldl -o ../../tools/abl/lib_bnf$v.so
done
ls ../../tools/abl/ | grep bnf
