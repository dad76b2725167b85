#!/bin/bash
# like ab_env.sh, for several values:  tools/ab_env2.sh VAR rounds v1 v2 v3 ... (a value "-" = variable unset)
VAR=$1; N=$2; shift 2
for i in $(seq 1 $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then E=""; else E="$VAR=$v"; fi
    ms=$(env $E python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra --train-only 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'])")
    echo "$VAR=$v  $ms"
  done
done
