#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace split of the batched model detector (tools/bench_model_detect.py [S] [bf16|f32]).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace -d gpurun_out/md_$tag -o md --output-format csv -- python3 tools/bench_model_detect.py $cfg 2>/dev/null | grep streams
  f=$(find gpurun_out/md_$tag -name '*kernel_trace.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "rp::" in r["Kernel_Name"]: d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in d.items(): print("   %-90s calls %d median %.3f ms" % (k[:90], len(v), statistics.median(v)))
PY
done
