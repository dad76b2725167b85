#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
one() { python bench.py --hrnet-width 32 --batch 32 --precision fp16 --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d.get('c_abi_calls_per_step'), d.get('loss_after'))"; }
for rep in 0 1 2; do
  for v in "$@"; do
    tag=${v%%:*}; envs=${v#*:}
    echo "rep$rep $tag [$envs]  $(env $envs LH_TUNE_CACHE=$PWD/gpurun_out/r6_hr_$tag.txt bash -c "$(declare -f one); one")" | tee -a gpurun_out/r6_hr_ab.txt
  done
done
