#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --templates 3 --template-len 126 2>/dev/null | grep '^{' | python3 -c "
import json,sys
x=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('t3: %.1f M  %.3f ms' % (x['value']/1e6, x['ms_per_step']), x['roofline']['kernels_ms'], x['roofline']['kernel'])"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys
x=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=x['roofline']; print('default: %.1f M  %.3f ms' % (x['value']/1e6, x['ms_per_step']), r['kernels_ms'], r['kernel'], 'frac %.3f exec %.3f mfma %.3f' % (r['frac'], r['executed_flop_frac'], r.get('mfma_f16_frac', 0)))"
