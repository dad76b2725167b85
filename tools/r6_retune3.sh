#!/bin/bash
# same-box A/B: the shipped database against the three-session ensemble
line() { python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e=d.get('extra',{})
print('train ms', d['ms_per_step'], 'infer img/s', d.get('infer_images_per_s'), 'hrnet ms', e.get('hrnet_w32_train_bs32',{}).get('ms_per_step'), 'c5 ms', e.get('r50_infer_384_bs256_fp16',{}).get('ms_per_step'), 'fp32 ms', e.get('r50_train_256_bs64_fp32',{}).get('ms_per_step'))"; }
for rep in 1 2 3; do
  echo "shipped   $(LH_TUNE_CACHE=0 line)" | tee -a gpurun_out/r6_retune_ab3.txt
  echo "ensemble  $(LH_TUNE_CACHE=0 LH_TUNE_DB=$PWD/tools/abl/tune_db_ensemble_full.txt line)" | tee -a gpurun_out/r6_retune_ab3.txt
done
