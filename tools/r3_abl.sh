#!/bin/bash
# Runs ON THE GPU BOX: timing ablations of mlp_stream_kernel (results wrong by design)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3abl; mkdir -p $O
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-pass-failed"
for a in ${ABLS:-0 5 6 7}; do
  touch rustpotter_amd/csrc/rp_mlp_stream.hip
  make -C rustpotter_amd/csrc -j8 CXXFLAGS="$BASE -DRP_STREAM_ABL=$a" > $O/make_$a.log 2>&1
  timeout 300 python3 bench.py --config C5 --steps 50 --warmup 5 --no-cpu-baseline > $O/abl_$a.json 2> $O/abl_$a.err
  python3 - <<PY
import json
try:
    j=json.loads(open("$O/abl_$a.json").read().strip().splitlines()[-1]); r=j["roofline"]
    print("ablation $a: %.4f ms  frac %.3f" % (r["avg_launch_ms"], r["frac"]))
except Exception as e: print("ablation $a failed", e)
PY
done
touch rustpotter_amd/csrc/rp_mlp_stream.hip; make -C rustpotter_amd/csrc -j8 > /dev/null 2>&1
