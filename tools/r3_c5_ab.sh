#!/bin/bash
# Runs ON THE GPU BOX: A/B of build variants of the wakeword-model kernels (rp_mlp_stream.hip, rp_mlp.hip) on ONE box, interleaved.
# Usage: r3_gx_ab.sh "<flags A>" "<flags B>" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c5ab; mkdir -p $O
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-pass-failed"
n=0
for flags in "$@"; do
  touch rustpotter_amd/csrc/rp_mlp_stream.hip rustpotter_amd/csrc/rp_mlp.hip
  make -C rustpotter_amd/csrc -j8 CXXFLAGS="$BASE $flags" > $O/make_$n.log 2>&1 || { tail -5 $O/make_$n.log; exit 1; }
  cp rustpotter_amd/librustpotter_hip.so $O/lib_$n.so
  n=$((n+1))
done
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    cp $O/lib_$i.so rustpotter_amd/librustpotter_hip.so
    for w in ${RP_AB_WORKLOADS:-"stream1:--mode_stream_--chunks-per-call_1" "k16:--streams_8192_--mfcc-size_16"}; do
      name=${w%%:*}; args=${w#*:}; args=${args//_/ }
      timeout 600 $B $args 2> $O/${name}_${i}_$rep.err | grep '^{' | tail -1 > $O/${name}_${i}_$rep.json
      python3 - <<PY
import json
j=json.loads(open("$O/${name}_${i}_$rep.json").read())
k=(j.get("roofline") or {}).get("kernels_ms") or j["config"].get("kernels_ms")
print("variant $i [$flags] rep $rep $name: %.1f M/s  step %.4f ms  kernels %s" % (j["value"]/1e6, j["ms_per_step"], k))
PY
    done
    i=$((i+1))
  done
done
cp $O/lib_$((n-1)).so rustpotter_amd/librustpotter_hip.so
rm -f $O/lib_*.so
