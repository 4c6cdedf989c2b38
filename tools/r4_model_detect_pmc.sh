#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over tools/bench_model_detect.py; prints per-kernel medians.  Usage: tools/r4_model_detect_pmc.sh [S] [bf16|f32]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/md_pmc_*
for c in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/md_pmc_$tag -o pmc -- python3 tools/bench_model_detect.py "$@" > gpurun_out/md_pmc_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, statistics
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/md_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rp::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/md_pmc_GRBM_GUI_ACTIVE/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rp::" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    if k in dur and statistics.median(dur[k]) < 200000: continue
    print(k, "median ns", statistics.median(dur[k]) if k in dur else None)
    for c, v in sorted(d.items()):
        print("    %-26s %.5g" % (c, statistics.median(v)))
PY
