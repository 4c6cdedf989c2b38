#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for cpc in 1 8; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stream$cpc -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --mode stream --chunks-per-call $cpc > gpurun_out/prof_stream$cpc.log 2>&1
grep '^{' gpurun_out/prof_stream$cpc.log | cut -c1-200
f=$(find gpurun_out/prof_stream$cpc -name '*kernel_stats.csv' | head -1); python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-60s calls %5s  avg %9.1f us  total %8.2f ms  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
done
