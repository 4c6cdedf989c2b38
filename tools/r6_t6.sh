#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_t6; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -200 > $O/gpu_tests.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | head -40
bash tools/prof.sh round 2>&1 | tail -60
