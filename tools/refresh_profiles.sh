#!/bin/bash
# Runs ON THE GPU BOX (gpurun): bench line, rocprofv3 kernel stats of the same command, and the two PMC passes.
# Everything lands in gpurun_out/refresh/; tools/refresh_profiles_post.sh (run in the build container) copies the
# summaries into profiles/.  The first bench run writes the autotuner's choices to a file (LH_TUNE_CACHE) that the
# profiled runs start from, so their kernel statistics hold the step's launches only, not the tuner's trials.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
export LH_TUNE_CACHE=$PWD/$O/tune_cache.txt
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r02 -- python3 bench.py --no-cpu-baseline --no-extra > $O/stats.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $O/pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_mfma -o m -- python3 tools/conv_bench.py 256 256 3 1 64 64 64 bf16 3 > $O/pmc_mfma.log 2>&1
timeout -k 10 300 python tools/layer_profile.py > $O/layers.txt 2>&1
find $O -name "*.csv" | head -20
