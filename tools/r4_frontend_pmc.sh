#!/bin/bash
# Runs ON THE GPU BOX: PMC passes (each its own run, kernel-trace only) over tools/bench_frontend.py; prints per-kernel medians.
# Usage: tools/r4_frontend_pmc.sh [bench_frontend args...]     -> gpurun_out/fe_pmc_*/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/fe_pmc_$tag -o pmc -- python3 tools/bench_frontend.py "$@" > gpurun_out/fe_pmc_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, statistics
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/fe_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rp::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/fe_pmc_GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rp::" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    print(k, "median ns", statistics.median(dur[k]) if k in dur else None)
    for c, v in sorted(d.items()):
        print("    %-26s %.5g" % (c, statistics.median(v)))
PY
