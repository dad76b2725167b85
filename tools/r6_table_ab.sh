#!/bin/bash
# round 6: table launches of the weight gradient -- tests, the tuning ladder of every table of the R50 step, A/B against per-layer launches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_wgrad_table.py -x -q > gpurun_out/r6_table_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r6_table_tests.log
tail -5 gpurun_out/r6_table_tests.log
LH_WGRAD_TABLE_LOG=1 timeout -k 10 900 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r6_table_ladder.txt 2>&1; echo "ladder rc=$?"
tail -3 gpurun_out/r6_table_ladder.txt | cut -c1-400
tools/ab_env.sh LH_WGRAD_TABLE 0 1 3 2>&1 | tee gpurun_out/r6_table_ab.txt
