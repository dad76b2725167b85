#!/bin/bash
# Runs ON THE GPU BOX (gpurun): measures the kernel choices of the benchmark configurations into gpurun_out/tune_db.txt.
# Copy the result (below its three comment lines) into lighthand_amd/tune_db_gfx950.txt in the build container.
set -e
cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/tune_db.txt
LH_TUNE_ITERS=20 LH_TUNE_DB=0 LH_TUNE_CACHE=$PWD/gpurun_out/tune_db.txt python bench.py --no-cpu-baseline --steps 5 > /dev/null
wc -l gpurun_out/tune_db.txt
