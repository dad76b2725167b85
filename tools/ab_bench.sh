#!/bin/bash
# A/B on ONE box: the product library against another build (default tools/abl/lib_base.so), alternating, training step +
# inference + C5.   usage: tools/ab_bench.sh [other.so] [rounds]
other=${1:-tools/abl/lib_base.so}; rounds=${2:-2}
line() { python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('extra',{})
print('$tag', 'train ms', d['ms_per_step'], 'infer img/s', d.get('infer_images_per_s'), 'hrnet ms', e.get('hrnet_w32_train_bs32',{}).get('ms_per_step'), 'c5 ms', e.get('r50_infer_384_bs256_fp16',{}).get('ms_per_step'))"; }
for i in $(seq $rounds); do
  tag=base; LH_LIB_PATH=$other line
  tag=new; line
done
