#!/bin/bash
# Runs ON THE GPU BOX: how the DTW kernel's time grows with the number of 32-window tiles around BASELINE config C2 (1 024 streams =
# 9 504 tiles = 3.09 rounds of the 3 072 resident waves): whole rounds (331 / 662 / 993 streams), C2 itself, and a few sizes beyond.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for s in ${SIZES:-166 331 662 993 1024 1324 2048 4096}; do
  python3 bench.py --streams $s --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']
tiles=$s*297/32.0
print('streams %5d  tiles %7.0f  rounds %.2f  step %.4f ms  mfcc %.4f  dtw %.4f  scan %.4f  M/s %.1f  dtw us per round %.1f' % ($s, tiles, tiles/3072, j['ms_per_step'], k['mfcc'], k['dtw'], k['scan'], j['value']/1e6, k['dtw']*1e3/(tiles/3072)))"
done
