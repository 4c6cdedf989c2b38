#!/bin/bash
# Runs ON THE GPU BOX: A/B of rp_dtw_mfma.hip build variants on ONE box, interleaved.  Usage: r3_dtwmfma_ab.sh "<flags A>" "<flags B>" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3dtwab; mkdir -p $O
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-pass-failed"
n=0
for flags in "$@"; do
  touch rustpotter_amd/csrc/rp_dtw_mfma.hip
  make -C rustpotter_amd/csrc -j8 CXXFLAGS="$BASE $flags" > $O/make_$n.log 2>&1 || { tail -5 $O/make_$n.log; exit 1; }
  cp rustpotter_amd/librustpotter_hip.so $O/lib_$n.so
  n=$((n+1))
done
for rep in 1 2 3; do
  i=0
  for flags in "$@"; do
    cp $O/lib_$i.so rustpotter_amd/librustpotter_hip.so
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/b_${i}_$rep.json 2> $O/b_${i}_$rep.err
    python - <<PY
import json
j=json.loads(open("$O/b_${i}_$rep.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("variant $i [$flags] rep $rep: mfcc %.4f dtw %.4f  step %.3f" % (r["kernels_ms"]["mfcc"], r["kernels_ms"]["dtw"], j["ms_per_step"]))
PY
    i=$((i+1))
  done
done
rm -f $O/lib_*.so
