#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of BASELINE config C2 -- per kernel the average duration, and the gaps between consecutive kernels of a step
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/trace_c2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 bench.py ${BENCH_ARGS:---config C2} --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "rp::" in r["Kernel_Name"]]
short = lambda n: n.split("(")[0].replace("void ", "").replace("rp::", "")[:60]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k, v in dur.items(): print("%-62s n=%4d  avg %8.2f us  min %8.2f" % (k, len(v), sum(v[len(v)//2:]) / len(v[len(v)//2:]) / 1e3, min(v) / 1e3))
for k, v in gap.items():
    if len(v) > 10: print("gap %-110s avg %7.2f us" % (k, sum(v[len(v)//2:]) / len(v[len(v)//2:]) / 1e3))
PY
