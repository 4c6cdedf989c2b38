#!/bin/bash
# Runs ON THE GPU BOX: ScoreMode::Max inside the matrix-core DTW kernel (default) against the aggregate pass (RP_DTW_NO_FUSED_MAX=1), interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3fused; mkdir -p $O
B="python3 bench.py --warmup 3 --no-cpu-baseline"
for rep in 1 2 3; do for nf in 1 0; do
  for w in "c3:--steps 5 --config C3" "c2:--steps 50 --config C2" "stream1:--steps 30 --mode stream --chunks-per-call 1" "t3:--steps 5 --templates 3 --template-len 126" "detect:--steps 5 --detect-only"; do
    name=${w%%:*}; args=${w#*:}
    if [ $nf = 1 ]; then export RP_DTW_NO_FUSED_MAX=1; else unset RP_DTW_NO_FUSED_MAX; fi
    timeout 600 $B $args 2> $O/${name}_${nf}_$rep.err | grep '^{' | tail -1 > $O/${name}_${nf}_$rep.json
    python3 - <<PY
import json
j=json.loads(open("$O/${name}_${nf}_$rep.json").read())
k=(j.get("roofline") or {}).get("kernels_ms") or j["config"].get("kernels_ms")
print("aggregate_pass=$nf rep $rep $name: %.1f M/s  step %.4f ms  kernels %s" % (j["value"]/1e6, j["ms_per_step"], k))
PY
  done; done; done
