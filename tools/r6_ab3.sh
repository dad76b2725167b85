#!/bin/bash
# round 6: stragglers in tables; the BN-backward gate's size cap re-measured on the round-6 step
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wgrad_table.py tests/test_gpu_runtime.py -x -q -k "table or lh_comm or c4_hrnet" > gpurun_out/r6_ab3_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6_ab3_tests.log
one() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d.get('c_abi_calls_per_step'))"; }
for rep in 1 2 3; do
  for v in 0 1; do echo "LH_WGRAD_TABLE_STRAGGLERS=$v  $(LH_WGRAD_TABLE_STRAGGLERS=$v one)" | tee -a gpurun_out/r6_ab3.txt; done
done
for rep in 1 2; do
  for mb in 0 9 40 1000; do echo "LH_BN_GATE_MAX_MB=$mb  $(LH_BN_GATE_MAX_MB=$mb one)" | tee -a gpurun_out/r6_ab3.txt; done
done
