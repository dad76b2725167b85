#!/bin/bash
# training-step A/B of several library builds on ONE box, alternating.  usage: tools/ab_train.sh rounds lib1.so lib2.so ...   ("product" = the in-tree build)
rounds=$1; shift
for i in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = product ]; then unset LH_LIB_PATH; else export LH_LIB_PATH=$lib; fi
    python bench.py --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['ms_per_step_median'], d.get('infer_images_per_s'))"
  done
done
