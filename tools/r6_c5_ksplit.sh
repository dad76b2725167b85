#!/bin/bash
# round 6: C5 (R50 384x384 bs256 fp16 inference) with fresh measurements, K-split forms offered or not
mkdir -p gpurun_out
for v in 0 1; do
  LH_KSPLIT_TILES=$v LH_TUNE_DB=0 LH_TUNE_ITERS=6 LH_TUNE_CACHE=$PWD/gpurun_out/r6_tune_c5_ks$v.txt python bench.py --infer-only --size 384 --batch 256 --precision fp16 --steps 5 --warmup 2 > /dev/null 2>&1
done
for rep in 1 2 3; do
  for v in 0 1; do
    ms=$(LH_KSPLIT_TILES=$v LH_TUNE_DB=0 LH_TUNE_CACHE=$PWD/gpurun_out/r6_tune_c5_ks$v.txt python bench.py --infer-only --size 384 --batch 256 --precision fp16 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "C5 fresh tuning, LH_KSPLIT_TILES=$v  $ms" | tee -a gpurun_out/r6_c5_ksplit.txt
  done
  ms=$(python bench.py --infer-only --size 384 --batch 256 --precision fp16 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
  echo "C5 shipped database  $ms" | tee -a gpurun_out/r6_c5_ksplit.txt
done
grep -c ", 3[234], 128))" gpurun_out/r6_tune_c5_ks1.txt
