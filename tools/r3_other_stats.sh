#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the other bench workloads (one text summary).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3other; mkdir -p $O; : > $O/summary.txt
run() {  # tag, bench args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/$tag.log 2>&1
  { echo "== python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $*"; grep '^{' $O/$tag.log | python3 -c "
import json,sys
x=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench line: %.4g %s, %.4f ms per step' % (x['value'], x['unit'], x['ms_per_step']))"
    f=$(find $O/$tag -name '*kernel_stats.csv' | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("   %-62s calls %4s  avg %10.1f us  %5.1f %%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  } >> $O/summary.txt
}
run c2 --config C2
run c4share --streams 8192 --templates 64
run t3 --templates 3 --template-len 126
run ragged5 --template-lens 108,96,90,93,102
run k16 --streams 8192 --mfcc-size 16
run k13 --streams 8192 --mfcc-size 13
run detect_only --detect-only
run gate04 --avg-gate --avg-threshold 0.4
run stream1 --mode stream --chunks-per-call 1
run stream8 --mode stream --chunks-per-call 8
cat $O/summary.txt
