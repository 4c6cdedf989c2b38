#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/sweep
SEED=23
timeout 4200 python3 tests/sweep_parity.py --seed $SEED --cases 500 --mfma-cases 1000 --ragged-cases 500 --api-cases 150 --live-multi-cases 300 --multi-cases 150 --model-cases 150 \
    --reset-cases 100 --rate-cases 60 --mfcc-cases 1000 --frontend-cases 40 --resample-cases 30 --builder-cases 20 --train-cases 8 --extreme-cases 120 2>&1 |
    grep -v "case [0-9]* ok\|amdgpu.ids" > gpurun_out/sweep/sweep_$SEED.txt
tail -30 gpurun_out/sweep/sweep_$SEED.txt
