#!/bin/bash
# round 6: a fresh measurement of every kernel choice of the benchmark configurations (20 launches per candidate), then old database vs new on the same box
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/tune_db_fresh.txt
LH_TUNE_ITERS=20 LH_TUNE_DB=0 LH_TUNE_CACHE=$PWD/gpurun_out/tune_db_fresh.txt timeout -k 10 1100 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/tune_db_fresh_bench.json 2> gpurun_out/tune_db_fresh.err; echo "retune rc=$?"
wc -l gpurun_out/tune_db_fresh.txt
line() { python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d.get('extra',{})
print('train ms', d['ms_per_step'], 'infer img/s', d.get('infer_images_per_s'), 'hrnet ms', e.get('hrnet_w32_train_bs32',{}).get('ms_per_step'), 'c5 ms', e.get('r50_infer_384_bs256_fp16',{}).get('ms_per_step'), 'fp32 ms', e.get('r50_train_256_bs64_fp32',{}).get('ms_per_step'))"; }
for rep in 1 2; do
  echo "shipped  $(LH_TUNE_CACHE=0 line)" | tee -a gpurun_out/r6_retune_ab.txt
  echo "fresh    $(LH_TUNE_CACHE=0 LH_TUNE_DB=$PWD/gpurun_out/tune_db_fresh.txt line)" | tee -a gpurun_out/r6_retune_ab.txt
done
