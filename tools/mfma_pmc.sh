#!/bin/bash
# MFMA-utilisation counters of the hot path's matrix kernels (tools/mfma_pmc.py): one --pmc pass for the SQ counters + kernel trace for
# the durations, then tools/mfma_pmc_table.py -> gpurun_out/mfma_pmc.json (committed as profiles/r05*_mfma_pmc.json, read by bench.py).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/mfma_pmc; rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python3 tools/mfma_pmc.py > $OUT/log1.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/t -o t --output-format csv -- python3 tools/mfma_pmc.py > $OUT/log2.txt 2>&1
python3 tools/mfma_pmc_table.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
