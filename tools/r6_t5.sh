#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_t5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dtw_f64.py tests/test_gpu_arithmetic.py tests/test_gpu_dtw_mfma.py -m gpu -q -x 2>&1 | tail -5
tools/ab.sh -r 3 -w "--steps 10 --warmup 3" -w "--mode mlp --mlp-precision f32" -- "" "-DRP_P3_NO_OVERLAP" 2>&1 | tail -14
