#!/bin/bash
# Runs ON THE GPU BOX: PMC passes over the DTW MFMA probe (counters only, no tracing besides the kernel trace).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3dtwmfma_pmc; mkdir -p $O; export TMPDIR=/tmp
read -r nw wg flags <<< "${1:-12 1 -fno-slp-vectorize -DRP_SCALAR_ADD=1}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DNWAVES=$nw -DWGS_PER_CU=$wg $flags tools/scratch/${PROBE:-dtw_mfma_probe2}.hip -o /tmp/dtw_probe_pmc || exit 1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH" \
           "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL"; do
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o pmc --output-format csv -- /tmp/dtw_probe_pmc ${ARGS:-100 8192 288} > $O/run$i.log 2>&1
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'dtw_mfma' not in r.get('Kernel_Name', ''): continue
    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in acc: print("%-34s per launch %.4g  (%d launches)" % (k, acc[k] / max(n[k], 1), n[k]))
PY
  i=$((i+1))
done
