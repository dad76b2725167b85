#!/bin/bash
# Runs ON THE GPU BOX (gpurun): bench line, rocprofv3 kernel stats of the same command, the PMC passes (HBM traffic, MFMA
# busy of the step's own kernels), per-layer tables, and the same for the two other single-GPU configurations of
# BASELINE.json (HRNet-W32 bs32 training = C4's per-GPU share, R50 384x384 bs256 fp16 inference = C5).
# Everything lands in gpurun_out/refresh/; tools/refresh_profiles_post.sh (run in the build container) copies the
# summaries into profiles/.  The first bench run writes the autotuner's choices to a file (LH_TUNE_CACHE) that the
# profiled runs start from, so their kernel statistics hold the step's launches only, not the tuner's trials.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
export LH_TUNE_CACHE=$PWD/$O/tune_cache.txt
R=${LH_ROUND:-r06}
LH_TUNE_ITERS=20 timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $R -- python3 bench.py --no-cpu-baseline --no-extra > $O/stats.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only > $O/pmc_fetch.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only > $O/pmc_write.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $O/pmc_mfma.log 2>&1
timeout -k 10 300 python tools/layer_profile.py > $O/layers.txt 2>&1
echo "[refresh] C4 share: HRNet-W32 bs32, fp16 + static loss scale (the timed dtype since round 4)"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_hrnet -o h -- python3 bench.py --hrnet-width 32 --batch 32 --precision fp16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > $O/stats_hrnet.log 2>&1
timeout -k 10 300 python tools/layer_profile.py --hrnet-width 32 --batch 32 --precision fp16 > $O/layers_hrnet.txt 2>&1
echo "[refresh] C5: R50 384x384 bs256 fp16 inference"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o c -- python3 bench.py --infer-only --size 384 --batch 256 --precision fp16 --steps 10 --warmup 3 > $O/stats_c5.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c5 -o f -- python3 bench.py --infer-only --size 384 --batch 256 --precision fp16 --steps 2 --warmup 1 > $O/pmc_fetch_c5.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c5 -o w -- python3 bench.py --infer-only --size 384 --batch 256 --precision fp16 --steps 2 --warmup 1 > $O/pmc_write_c5.log 2>&1
timeout -k 10 300 python tools/layer_profile.py --infer --size 384 --batch 256 --precision fp16 > $O/layers_c5.txt 2>&1
find $O -name "*.csv" | head -30
