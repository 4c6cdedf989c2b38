#!/bin/bash
# Runs ON THE GPU BOX: a longer run of the sweep families that touch this round's kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3sweep
timeout 5400 python tests/sweep_parity.py --seed ${RP_SWEEP_SEED:-11} --cases 1000 --mfma-cases 2500 --api-cases 300 --live-multi-cases 600 --multi-cases 300 --reset-cases 200 --rate-cases 100 --extreme-cases 100 2>&1 | grep -v "case [0-9]* ok\|amdgpu.ids" > gpurun_out/r3sweep/sweep_long.txt
tail -12 gpurun_out/r3sweep/sweep_long.txt
