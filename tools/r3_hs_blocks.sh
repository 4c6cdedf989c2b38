#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for b in 1536 768 3072 6144 16384; do
RP_MFCC_HS_BLOCKS=$b python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mode stream --chunks-per-call 1 2>/dev/null | grep '^{' | python3 -c "
import json,sys
x=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks $b rep $rep: %.1f M  %.4f ms' % (x['value']/1e6, x['ms_per_step']))"
done; done
