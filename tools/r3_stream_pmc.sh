#!/bin/bash
# Runs ON THE GPU BOX: PMC passes of bench.py --mode stream (the live-stream launch of the matrix-core DTW kernel, frames from global memory)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/streampmc
O=gpurun_out/streampmc
run() { tag=$1; shift; ctrs=$1; shift
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$tag -o pmc -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --mode stream --chunks-per-call 1 "$@" > $O/$tag.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/$tag/**/*counter_collection.csv", recursive=True)
if not f:
    print("$tag: no counter csv"); print(open("$O/$tag.log").read()[-1500:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "dtw" in k or "mfcc" in k:
        print("$tag", k, {c: "%.4g" % (sum(v[-4:]) / len(v[-4:])) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
}
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "$@"
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "$@"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "$@"
run grbm "GRBM_GUI_ACTIVE" "$@"
