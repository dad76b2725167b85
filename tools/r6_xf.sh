#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
one() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['c_abi_calls_per_step'], d['loss_after'])"; }
V=("K:LH_BN_APPLY_IN=0" "M30:LH_BN_APPLY_IN_MIN_MB=30" "M100:LH_BN_APPLY_IN_MIN_MB=100" "B:LH_X=1")
for rep in 0 1 2; do
  for v in "${V[@]}"; do
    tag=${v%%:*}; envs=${v#*:}
    echo "rep$rep $tag [$envs]  $(env $envs LH_TUNE_ITERS=10 LH_TUNE_CACHE=$PWD/gpurun_out/r6_xf_$tag.txt bash -c "$(declare -f one); one")" | tee -a gpurun_out/r6_xf_ab.txt
  done
done
