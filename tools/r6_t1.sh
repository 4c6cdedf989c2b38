#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_t1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -150 > $O/gpu_tests.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | head -80
tools/ab.sh -r 2 -w "--steps 10 --warmup 3" -- "" "-DRP_P3_GAP=2" "-DRP_P3_GAP=0" 2>&1 | tail -12
