#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace split of rp_frontend_batch at C3 size for the given "TILE TURNS GAIN BP FMT" settings.
# Usage: tools/r4_frontend_prof.sh "0 1 1 1 i16" ...   (TILE / TURNS are exported as RP_FRONTEND_TILE / RP_FRONTEND_TURNS: only TILE = 128 still selects another form)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  tag=$(echo $cfg | tr ' ' '_')
  export RP_FRONTEND_TILE=$1 RP_FRONTEND_TURNS=$2
  rocprofv3 --kernel-trace --stats -d gpurun_out/fe_$tag -o fe --output-format csv -- python3 tools/bench_frontend.py 65536 $3 $4 $5 2>/dev/null | grep frontend | sed "s/^/tile=$1 turns=$2 /"
  f=$(find gpurun_out/fe_$tag -name '*kernel_trace.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "rp::" in r["Kernel_Name"]: d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in d.items(): print("   %-76s calls %d median %.3f ms min %.3f" % (k[:76], len(v), statistics.median(v), min(v)))
PY
done
