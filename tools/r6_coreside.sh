#!/bin/bash
# round 6: can the deferred weight-gradient tables run UNDER the backward chain?  The 8-wave 256x256 table kernel holds 254 VGPRs x 2 waves
# per SIMD: nothing of the chain fits beside it (the chain's reduce pass: 55 -> 423 us while the table runs).  The 128x128 class (146 VGPRs,
# 4 waves, 64 KB LDS) leaves room for the chain's BatchNorm passes.  Step A/B: table classes x layers per deferred group.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_coreside.txt; : > $O
one() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'])"; }
run() { # tag, env...
  local tag=$1; shift
  echo "$tag  $(env "$@" LH_TUNE_CACHE=$PWD/gpurun_out/r6_cs_$tag.txt bash -c "$(declare -f one); one")" | tee -a $O
}
V=("A:LH_X=0" "B:LH_WGRAD_TABLE_BIG=0" "C:LH_WGRAD_TABLE_BIG=0 LH_WGRAD_GROUP=12" "D:LH_WGRAD_TABLE_BIG=0 LH_WGRAD_GROUP=24" "E:LH_WGRAD_GROUP=12" "F:LH_WGRAD_TABLE_BIG=0 LH_WGRAD_GROUP=18 LH_WGRAD_LANES=3")
for rep in 0 1 2; do
  for v in "${V[@]}"; do
    tag=${v%%:*}; envs=${v#*:}
    [ $rep = 0 ] && tag2="warm_$tag" || tag2=$tag
    echo -n "rep$rep [$envs] " | tee -a $O
    run $tag $envs
  done
done
# timeline of variant B
env LH_WGRAD_TABLE_BIG=0 LH_TUNE_CACHE=$PWD/gpurun_out/r6_cs_B.txt timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6_cs_trace -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra --train-only > gpurun_out/r6_cs_trace.log 2>&1
f=$(find gpurun_out/r6_cs_trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f gpurun_out/r6_cs_timeline_B.txt
cp $f gpurun_out/r6_cs_trace_B.csv; rm -rf gpurun_out/r6_cs_trace
