#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass (own run, kernel-trace only) of bench.py.
# Usage: tools/profile_pmc.sh <tag> "<COUNTER1 COUNTER2 ...>" [bench args...]
set -u
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc_${TAG} -o pmc -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/pmc_${TAG}.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_${TAG}/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); print(open("gpurun_out/pmc_${TAG}.log").read()[-2000:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
