#!/bin/bash
# round 6: BatchNorm-backward gate on the pointwise / direct 3x3 kernels and for residual tails -- tests, then the step with / without
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -k "gate or pointwise or every_conv or direct3x3" > gpurun_out/r6_gate_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r6_gate_tests.log
[ $rc = 0 ] || exit 1
O=gpurun_out/r6_gate_ab.txt; : > $O
one() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>gpurun_out/r6_gate_err_$TAG.txt | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['c_abi_calls_per_step'], d['loss_after'])"; }
V=("A:LH_BN_GATE_PW=0 LH_BN_GATE_TAIL=0" "B:LH_X=1" "J:LH_BN_GATE_TAIL2=0")
for rep in 0 1 2; do
  for v in "${V[@]}"; do
    tag=${v%%:*}; envs=${v#*:}
    echo "rep$rep $tag [$envs]  $(env $envs TAG=$tag LH_TUNE_LOG=$([ $rep = 0 ] && echo 1) LH_TUNE_CACHE=$PWD/gpurun_out/r6_gate_$tag.txt bash -c "$(declare -f one); one")" | tee -a $O
  done
done
python tools/gate_list.py r50 64 256 > gpurun_out/r6_gate_list.txt 2>&1
