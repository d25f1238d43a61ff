cd $GRAFT_REPO_ROOT
bash tools/run_r5i.sh r05i
python -m pytest tests/test_gpu_r5.py -q 2>&1 | tail -3
HVLM_LIB=$PWD/handsonvlm-release_amd/lib/libhvlm_hip_dev.so python -m pytest tests -m gpu -q -k "gemm or clip or layernorm_fold" 2>&1 | tail -3 > gpurun_out/r05j_dev_invariant.txt; cat gpurun_out/r05j_dev_invariant.txt
