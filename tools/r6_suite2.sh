#!/bin/bash
# round 6: the whole GPU suite after the engine split; then the table choices of the benchmark configurations for the shipped database
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r6_suite2.log 2>&1; echo "suite rc=$?"
grep -E "FAILED|ERROR|passed|failed" gpurun_out/r6_suite2.log | tail -20
rm -f gpurun_out/tune_new.txt
LH_TUNE_ITERS=20 LH_TUNE_CACHE=$PWD/gpurun_out/tune_new.txt timeout -k 10 900 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/tune_new_bench.json 2> gpurun_out/tune_new_bench.err; echo "tune rc=$?"
wc -l gpurun_out/tune_new.txt; grep -c "'wt'" gpurun_out/tune_new.txt
