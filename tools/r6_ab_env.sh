#!/bin/bash
# usage: r6_ab_env.sh "<tests -k expr or empty>" "TAG:ENV=.. ENV=.." ...   -- tests, then alternating step timings of the variants (3 rounds)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K="$1"; shift
if [ -n "$K" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_runtime.py -x -q -k "$K" > gpurun_out/r6_ab_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r6_ab_tests.log
  [ $rc = 0 ] || exit 1
fi
one() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['c_abi_calls_per_step'], d['loss_after'])"; }
for rep in 0 1 2; do
  for v in "$@"; do
    tag=${v%%:*}; envs=${v#*:}
    echo "rep$rep $tag [$envs]  $(env $envs LH_TUNE_CACHE=$PWD/gpurun_out/r6_ab_$tag.txt bash -c "$(declare -f one); one")" | tee -a gpurun_out/r6_ab_env.txt
  done
done
