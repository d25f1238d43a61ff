cd $GRAFT_REPO_ROOT
python -m pytest "tests/test_gpu_r5.py" -q -x -k "clip_attn_asm" 2>&1 | tail -30 > gpurun_out/r05h_clip_test.log
python -m pytest "tests/test_gpu_fulldepth_r4.py" -q -k "test_a_ or test_b_" 2>&1 | tail -60 > gpurun_out/r05h_testb.log
timeout 300 python3 tools/attn_asm_check.py --time > gpurun_out/r05h_krows_check.log 2>&1
timeout 300 python3 tools/attn_asm_prof.py > gpurun_out/r05h_krows_prof.log 2>&1
tail -5 gpurun_out/r05h_clip_test.log; tail -5 gpurun_out/r05h_testb.log; tail -8 gpurun_out/r05h_krows_check.log; cat gpurun_out/r05h_krows_prof.log
