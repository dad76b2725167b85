#!/bin/bash
# round 6: measure the gate-tagged launches for the shipped database (20 launches per candidate), then kernel statistics of the step
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LH_TUNE_CACHE=$PWD/gpurun_out/r6_gate_tune.txt
rm -f $LH_TUNE_CACHE
LH_TUNE_ITERS=20 LH_TUNE_TIMES=$PWD/gpurun_out/r6_gate_times.txt timeout -k 10 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra --no-roofline > gpurun_out/r6_gate_tune_bench.json 2> gpurun_out/r6_gate_tune.err
tail -c 300 gpurun_out/r6_gate_tune_bench.json
rm -rf gpurun_out/r6_gate_stats
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_gate_stats -o g -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra --no-roofline --train-only > gpurun_out/r6_gate_stats.log 2>&1
f=$(find gpurun_out/r6_gate_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r6_gate_kernel_stats.csv
t=$(find gpurun_out/r6_gate_stats -name "*kernel_trace.csv" | head -1); python tools/step_timeline.py $t gpurun_out/r6_gate_timeline.txt; cp $t gpurun_out/r6_gate_trace.csv
rm -rf gpurun_out/r6_gate_stats
