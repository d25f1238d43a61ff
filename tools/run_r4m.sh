#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_next_rows.py tests/test_gpu_ops.py -q -m gpu 2>&1 | tail -6
