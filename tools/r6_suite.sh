#!/bin/bash
# round 6: the whole GPU suite (no -x: list everything the table launches changed), HRNet / C5 A-B of the table launches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -v -p no:cacheprovider > gpurun_out/r6_suite.log 2>&1; echo "suite rc=$?"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/r6_suite.log | tail -40
for v in ; do
  ms=$(LH_WGRAD_TABLE=$v python bench.py --hrnet-width 32 --batch 32 --precision fp16 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d.get('c_abi_calls_per_step'))")
  echo "HRNet-W32 bs32 fp16 LH_WGRAD_TABLE=$v  $ms" | tee -a gpurun_out/r6_hrnet_table.txt
done
