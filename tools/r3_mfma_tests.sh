#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3mfmatests; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dtw_mfma.py -m gpu -q 2>&1 | tail -6
timeout 2400 python tests/sweep_parity.py --cases 0 --api-cases 0 --mfma-cases ${MFMA_CASES:-300} 2>&1 | grep -v "^  \|case [0-9]* ok\|amdgpu.ids" | tail -12 | tee $O/sweep.txt
