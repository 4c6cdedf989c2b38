#!/bin/bash
# Runs ON THE GPU BOX: the randomised sweeps of round 3 (new family + a re-run of the others on the round's kernels)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3sweep
timeout 3000 python tests/sweep_parity.py --cases 300 --mfma-cases 300 --api-cases 150 --live-multi-cases 400 --multi-cases 150 --model-cases 120 --reset-cases 100 --rate-cases 60 \
   --mfcc-cases 1500 --frontend-cases 60 --resample-cases 40 --builder-cases 30 --train-cases 10 --extreme-cases 60 2>&1 | grep -v "case [0-9]* ok\|amdgpu.ids" > gpurun_out/r3sweep/sweep.txt
tail -20 gpurun_out/r3sweep/sweep.txt
