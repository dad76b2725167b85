#!/bin/bash
# like ab_env2.sh with extra bench.py flags:  tools/ab_env3.sh "FLAGS" VAR rounds v1 v2 ...   ("-" = variable unset)
FLAGS=$1; VAR=$2; N=$3; shift 3
for i in $(seq 1 $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then E=""; else E="$VAR=$v"; fi
    ms=$(env $E python bench.py --no-cpu-baseline --no-roofline --no-extra --train-only $FLAGS 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'])")
    echo "$VAR=$v  $ms"
  done
done
