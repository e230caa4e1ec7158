#!/bin/bash
# Collect HBM-traffic and cache counters for one bench workload, one --pmc pass each
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/pmc_collect.sh <tag> <bench args...>
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
rm -rf $out
mkdir -p $out
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 bench.py --no-cpu-baseline --no-extras --no-graph --steps 60 --warmup 10 "$@" > $out/pass$i.json 2> $out/pass$i.err || echo "pass $i failed"
  echo "pass $i done: $ctrs"
done
