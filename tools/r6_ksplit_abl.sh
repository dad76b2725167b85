#!/bin/bash
# round 6: what binds the 128 x 128 tile on the stage-3 3x3 (256 -> 256 at 16 x 16, batch 64): the dense-wave form (8 waves of 64 x 32) and
# the K-split wave-pair form (8 waves as 4 pairs of 64 x 64) with parts of the stage loop removed (LH_ABL bits: 1 MFMAs, 2 fragment reads, 4 LDS-DMA)
O=gpurun_out/r6_ksplit_ablation.txt
rm -f $O
for cfg in 128,128,23,128 128,128,33,128; do
  for v in full 1 2 4 3 6 5; do
    lib=""; [ $v != full ] && lib=$PWD/tools/abl/lib_abl$v.so
    t=$(LH_LIB_PATH=$lib LH_FORCE_CFG=$cfg LH_WGRAD_TABLE=0 LH_TUNE_CACHE=0 python tools/conv_bench.py 256 256 3 1 64 16 16 bf16 200 2>/dev/null | grep -E "conv fwd|conv dgrad" | awk '{printf "%s %s us  ", $2, $(NF-3)}')
    echo "cfg $cfg  abl $v :  $t" | tee -a $O
  done
done
