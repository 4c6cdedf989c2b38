#!/bin/bash
# Runs ON THE GPU BOX: C5 on the stream kernel -- parity tests, ring depths, then the profiles/ evidence (kernel-trace stats + PMC passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c5c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "mlp or model or c5" > $O/tests.log 2>&1; tail -3 $O/tests.log
for p in bf16 f32; do
  for d in 2 1; do
    RP_MLP_STREAM_DEPTH=$d timeout 600 python3 bench.py --mode mlp --mlp-precision $p --steps 50 --warmup 5 --no-cpu-baseline > $O/c5_${p}_d$d.json 2> $O/c5_${p}_d$d.err
    python3 - <<PY
import json
j=json.loads(open("$O/c5_${p}_d$d.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("$p depth $d: %.1f M rows/s  %.4f ms  frac %.3f  %s" % (j["value"]/1e6, r["avg_launch_ms"], r["frac"], r["kernel"]))
PY
  done
done
timeout 600 python3 bench.py --config C5 --steps 50 --warmup 5 > $O/c5_bf16.json 2> $O/c5_bf16.err; tail -c 700 $O/c5_bf16.json
timeout 600 python3 bench.py --config C5 --mlp-precision f32 --steps 50 --warmup 5 > $O/c5_f32.json 2> $O/c5_f32.err
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 bench.py --config C5 --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_c5.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -o pmc -- python3 bench.py --config C5 --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/pmc_*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "mlp" in r["Kernel_Name"]: a[r["Kernel_Name"][:32]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in a.items(): print(d.split("/")[-2], k, {c: "%.5g" % (sum(x)/len(x)) for c, x in v.items()})
PY
