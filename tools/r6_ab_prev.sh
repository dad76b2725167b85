#!/bin/bash
# same-box A/B of the product library against the previous commit's build (tools/abl/lib_prev.so): the removals must not move the kernels
one() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d.get('infer_images_per_s'))"; }
for rep in 1 2 3; do
  echo "prev  $(LH_LIB_PATH=$PWD/tools/abl/lib_prev.so one)" | tee -a gpurun_out/r6_ab_prev.txt
  echo "new   $(one)" | tee -a gpurun_out/r6_ab_prev.txt
done
