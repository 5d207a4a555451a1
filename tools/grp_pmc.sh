#!/bin/bash
# (development) hardware counters of lsd_rg_grp: tools/grp_pmc.sh frames mode wpb   (run on the GPU box)
F=${1:-2048}; MODE=${2:-grp}; W=${3:-1}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r04a; mkdir -p $out
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_FLAT" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_DCACHE_REQ"; do
  i=$((i+1))
  WPBS=$W timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -o res -- python $R/tools/lsd_grp_check.py $F 32 $MODE > $out/pmc_run_$i.log 2>&1
  echo "== $set"
  python $R/tools/pmc_summary.py /tmp/pmc_$i lsd_rg_grp 2>&1 | tail -8
done
