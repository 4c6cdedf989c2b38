#!/bin/bash
# Runs ON THE GPU BOX: how many tiles a wave of the matrix-core DTW kernels takes by index before it uses the atomic counter
# (RP_MFMA_STATIC_ROUNDS: 0 = all from the counter, 1 = the first, unset = the host's rule), interleaved on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3sched; mkdir -p $O
B="python3 bench.py --warmup 3 --no-cpu-baseline"
for rep in 1 2; do for sr in 0 1 auto; do
  for w in "stream1:--steps 20 --mode stream --chunks-per-call 1" "stream2:--steps 20 --mode stream --chunks-per-call 2" "stream8:--steps 20 --mode stream --chunks-per-call 8" "c2:--steps 50 --config C2" "c3:--steps 5 --config C3" "k16:--steps 10 --streams 8192 --mfcc-size 16" "s2048:--steps 50 --streams 2048" "s4096:--steps 30 --streams 4096"; do
    name=${w%%:*}; args=${w#*:}
    if [ $sr = auto ]; then unset RP_MFMA_STATIC_ROUNDS; else export RP_MFMA_STATIC_ROUNDS=$sr; fi
    timeout 600 $B $args 2> $O/${name}_${sr}_$rep.err | grep '^{' | tail -1 > $O/${name}_${sr}_$rep.json
    python3 - <<PY
import json
j=json.loads(open("$O/${name}_${sr}_$rep.json").read())
k=(j.get("roofline") or {}).get("kernels_ms") or j["config"].get("kernels_ms")
print("static_rounds=$sr rep $rep $name: %.1f M/s  step %.4f ms  dtw %s" % (j["value"]/1e6, j["ms_per_step"], k["dtw"]))
PY
  done; done; done
