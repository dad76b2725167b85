#!/bin/bash
# round 6: merged small-channel class; group size of the deferred weight gradients with table launches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_wgrad_table.py -x -q > gpurun_out/r6_table_tests2.log 2>&1; echo "tests rc=$?"
LH_WGRAD_TABLE_LOG=1 timeout -k 10 900 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r6_table_ladder2.txt 2>&1; echo "ladder rc=$?"
tail -1 gpurun_out/r6_table_ladder2.txt | cut -c1-300
for g in 36 57 24 44; do
  for rep in 1 2; do
    ms=$(LH_WGRAD_GROUP=$g python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d.get('c_abi_calls_per_step'))")
    echo "LH_WGRAD_GROUP=$g  $ms" | tee -a gpurun_out/r6_table_groups.txt
  done
done
for l in 1 2 3; do
  ms=$(LH_WGRAD_LANES=$l python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'])")
  echo "LH_WGRAD_LANES=$l  $ms" | tee -a gpurun_out/r6_table_groups.txt
done
