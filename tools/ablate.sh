#!/bin/bash
# Builds debug variants of the library with parts of the igemm ring kernel's K loop removed (see LH_ABL in
# igemm_ring.hip) into tools/abl/, for timing experiments only:  LH_LIB_PATH=tools/abl/lib_abl1.so python ...
set -e
cd "$(dirname "$0")/../lighthand_amd/csrc"
mkdir -p ../../tools/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DLH_ABL=$v -c igemm_ring.hip -o /tmp/igemm_ring_abl$v.o
  /opt/rocm/bin/hipcc $FLAGS -DLH_ABL=$v -c wgrad.hip -o /tmp/wgrad_abl$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o /tmp/igemm_ring_abl$v.o /tmp/wgrad_abl$v.o bn.o misc.o -o ../../tools/abl/lib_abl$v.so
done
