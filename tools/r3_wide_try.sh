#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | head -20
for m in 1 0; do for k in 16 13; do
RP_DTW_MFMA=$m python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 8192 --mfcc-size $k 2>/dev/null | grep '^{' | python3 -c "
import json,sys
x=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=x.get('roofline') or {}; print('mfma=$m K=$k: %.1f M  %.4f ms' % (x['value']/1e6, x['ms_per_step']), r.get('kernels_ms'))"
done; done
