#!/bin/bash
# round 6: XCD-aware group placement of the table's work items -- tests, fresh ladders, step A/B, fabric bytes of the table launches
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_wgrad_table.py -x -q > gpurun_out/r6_xcd_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r6_xcd_tests.log
for v in 1 0; do
  LH_WGRAD_TABLE_XCD=$v LH_WGRAD_TABLE_LOG=1 LH_TUNE_DB=$PWD/gpurun_out/none.txt LH_TUNE_CACHE=$PWD/gpurun_out/r6_tune_xcd$v.txt timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r6_xcd_ladder$v.txt 2>&1
done
one() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['ms_per_step_median'])"; }
for rep in 1 2 3; do
  for v in 0 1; do echo "LH_WGRAD_TABLE_XCD=$v  $(LH_WGRAD_TABLE_XCD=$v LH_TUNE_DB=$PWD/gpurun_out/none.txt LH_TUNE_CACHE=$PWD/gpurun_out/r6_tune_xcd$v.txt one)" | tee -a gpurun_out/r6_xcd_ab.txt; done
done
for v in 0 1; do
  LH_WGRAD_TABLE_XCD=$v LH_TUNE_DB=$PWD/gpurun_out/none.txt LH_TUNE_CACHE=$PWD/gpurun_out/r6_tune_xcd$v.txt timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r6_xcd_pmc$v -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-only > gpurun_out/r6_xcd_pmc$v.log 2>&1
  python - <<PY
import csv,collections,glob
f=glob.glob("gpurun_out/r6_xcd_pmc$v/**/*counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda:[0.0,0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="FETCH_SIZE" and "wgrad" in r["Kernel_Name"]:
        a=acc[r["Kernel_Name"][:70]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,(v,n) in sorted(acc.items(), key=lambda kv:-kv[1][0]): print("XCD=$v", k, n, "launches", round(2*v*1024/n/1e6,1), "MB read per launch")
PY
done
