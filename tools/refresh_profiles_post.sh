#!/bin/bash
# Build-container side of tools/refresh_profiles.sh: copy the judged summaries into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/refresh
cp $O/bench.json profiles/r01_bench.json
cp "$(find $O/stats -name '*kernel_stats.csv' | head -1)" profiles/r01_bench_kernel_stats.csv
cp $O/layers.txt profiles/r01_layers.txt
python tools/pmc_traffic.py "$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)" \
    "$(find $O/pmc_write -name '*counter_collection.csv' | head -1)" profiles/r01_pmc_hbm_traffic.txt profiles/r01_pmc_traffic.json
ls -la profiles/
