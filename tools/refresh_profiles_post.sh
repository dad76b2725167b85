#!/bin/bash
# Build-container side of tools/refresh_profiles.sh: copy the judged summaries into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/refresh
cp $O/bench.json profiles/r02_bench.json
cp "$(find $O/stats -name '*kernel_stats.csv' | head -1)" profiles/r02_bench_kernel_stats.csv
cp $O/layers.txt profiles/r02_layers.txt
python tools/pmc_traffic.py "$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1)" \
    "$(find $O/pmc_write -name '*counter_collection.csv' | head -1)" profiles/r02_pmc_hbm_traffic.txt profiles/r02_pmc_traffic.json
python tools/pmc_mfma.py "$(find $O/pmc_mfma -name '*counter_collection.csv' | head -1)" profiles/r02_pmc_mfma.txt > /dev/null
ls -la profiles/
