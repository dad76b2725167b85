#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LH_TUNE_CACHE=$PWD/gpurun_out/r6_hr_gate_tune.txt
rm -f $LH_TUNE_CACHE
for prec in fp16 bf16; do
  LH_TUNE_ITERS=20 python bench.py --hrnet-width 32 --batch 32 --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | tail -c 200
done
wc -l $LH_TUNE_CACHE
