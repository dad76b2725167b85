#!/bin/bash
# Runs ON THE GPU BOX: hardware-counter passes (rocprofv3 --pmc, one pass per counter group) over ONE convolution launch
# under the listed kernel configurations, then the report (tools/pmc_conv.py).
# usage: tools/pmc_conv.sh <name> <GFLOP per launch> CIN COUT K STRIDE N H W TRANSPOSED prec "cfg;cfg;..."
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=$1; GF=$2; shift 2
O=gpurun_out/$N
rm -rf $O && mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -o p -- python3 tools/pmc_conv.py run "$@" > $O/p$i.log 2>&1
  echo "pass $i ($grp): rc $?" >> $O/passes.log
done
python tools/pmc_conv.py report $O/report.txt $GF $(find $O -name '*counter_collection.csv' | sort) > /dev/null 2>> $O/passes.log
cat $O/passes.log
