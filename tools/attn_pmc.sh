#!/bin/bash
# PMC passes over the decoder attention kernels (tools/attn_bench.py): issue / wait / LDS / MFMA counters, one pass per group.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/attn_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python3 tools/attn_bench.py > $OUT/log$i.txt 2>&1
done
python3 tools/pmc_sum.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
