#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_dtw_f64.py tests/test_gpu_arithmetic.py tests/test_gpu_dtw_mfma.py tests/test_gpu_dtw_group.py -m gpu -q -x 2>&1 | tail -3
tools/ab.sh -r 3 -w "--steps 10 --warmup 3" -w "--steps 10 --warmup 3 --arith fast_split" -- "" "-DRP_AREF_EARLY=0" "-DRP_MFMA3_PITCH=512" 2>&1 | tail -20
