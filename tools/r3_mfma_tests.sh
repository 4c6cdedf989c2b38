#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3mfmatests; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dtw_mfma.py -m gpu -q 2>&1 | tail -30
timeout 2400 python tests/sweep_parity.py --cases 400 --api-cases 100 --live-multi-cases 150 --multi-cases 100 --reset-cases 60 --extreme-cases 40 2>&1 | grep -v "^  " | tail -12 | tee $O/sweep.txt
