#!/bin/bash
# A/B of one environment switch on the training step, alternating runs inside ONE gpurun call (boxes differ by +-1.7 %):
#   tools/ab_env.sh VAR A B [rounds] [extra bench.py flags...]   ->  ms per step of each run
VAR=$1; A=$2; B=$3; N=${4:-3}; shift 4
for i in $(seq 1 $N); do
  for v in "$A" "$B"; do
    ms=$(env $VAR=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'])")
    echo "$VAR=$v  $ms"
  done
done
