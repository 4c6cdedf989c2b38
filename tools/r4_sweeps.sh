#!/bin/bash
# Runs ON THE GPU BOX: the randomised sweeps on round 4's kernels (every family; the matrix-core family now draws score_ref from
# 0.05..0.3 and the extreme-parameter family compares at 1e-5 from score_ref 0.05 up).  Usage: tools/r4_sweeps.sh [seed]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4sweep
timeout 5000 python tests/sweep_parity.py --seed ${1:-17} --cases 600 --mfma-cases 1500 --api-cases 200 --live-multi-cases 400 --multi-cases 200 --model-cases 200 --reset-cases 150 --rate-cases 80 \
   --mfcc-cases 1500 --frontend-cases 60 --resample-cases 40 --builder-cases 30 --train-cases 10 --extreme-cases 150 2>&1 | grep -v "case [0-9]* ok\|amdgpu.ids" > gpurun_out/r4sweep/sweep_${1:-17}.txt
tail -25 gpurun_out/r4sweep/sweep_${1:-17}.txt
