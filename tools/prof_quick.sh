#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel statistics of a short bench run into gpurun_out/<name>/ (tuner launches excluded
# through a warm LH_TUNE_CACHE).  usage: tools/prof_quick.sh <name> [bench args]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=$1; shift
O=gpurun_out/$N
rm -rf $O && mkdir -p $O
export LH_TUNE_CACHE=$PWD/$O/tune_cache.txt
python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 10 "$@" > $O/warm.json 2> $O/warm.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o q -- python3 bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 "$@" > $O/stats.log 2>&1
cut -c1-200 $O/warm.json
