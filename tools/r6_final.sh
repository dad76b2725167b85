#!/bin/bash
# round 6: the whole GPU suite + smoke + the default bench on the final tree
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r6_final_suite.log 2>&1; echo "suite rc=$?"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/r6_final_suite.log | tail -12
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6_final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r6_final_smoke.log
for rep in 1 2; do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step', d['ms_per_step'], d['ms_per_step_median'], d.get('infer_images_per_s'))"
done
python bench.py --infer-only --size 384 --batch 256 --precision fp16 --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('C5', d['ms_per_step'], d['value'])"
python bench.py --hrnet-width 32 --batch 32 --precision fp16 --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('HRNet', d['ms_per_step'], d['value'], d.get('c_abi_calls_per_step'))"
