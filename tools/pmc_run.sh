#!/bin/bash
# PMC passes (one counter group per rocprofv3 run; never combined with trace domains other than --kernel-trace) over a
# small target command.  usage: tools/pmc_run.sh <outdir> <command...>
mkdir -p $1; out=$(cd $1 && pwd); shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1 || echo "group $i ($grp) failed"
done
