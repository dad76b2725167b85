#!/bin/bash
# Debug builds of the fused inference bottleneck with parts removed (LH_BNK_ABL bits, bottleneck_infer_kernel.h) into tools/abl/,
# for timing experiments only:  LH_LIB_PATH=tools/abl/lib_bnk8.so python bench.py --infer-only ...     usage: tools/ablate_bneck.sh 1 2 8 ...
set -e
cd "$(dirname "$0")/../lighthand_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DLH_BNK_ABL=$v -c bottleneck_infer.hip -o /tmp/bottleneck_infer_abl$v.o
  OTHERS=$(ls *.o | grep -v "^bottleneck_infer.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/bottleneck_infer_abl$v.o -ldl -o ../../tools/abl/lib_bnk$v.so
done
ls ../../tools/abl/ | grep bnk
