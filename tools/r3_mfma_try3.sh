#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for cfg in "--config C2 --steps 50" "" "--mode stream --chunks-per-call 1" "--streams 8192 --templates 64"; do
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $cfg 2>/dev/null | grep '^{' | python3 -c "
import json,sys
x=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=x.get('roofline') or {}; print('$cfg: %.1f M  %.4f ms' % (x['value']/1e6, x['ms_per_step']), r.get('kernels_ms'))"
done
