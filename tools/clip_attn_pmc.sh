#!/bin/bash
# PMC passes over the CLIP resident attention kernel (tools/clip_attn_bench.py)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clip_attn_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/p$i -o p$i --output-format csv -- python3 tools/clip_attn_bench.py > $OUT/log$i.txt 2>&1
done
python3 tools/pmc_sum.py $OUT | grep resident > $OUT/summary.txt
cat $OUT/summary.txt; tail -3 $OUT/log3.txt
