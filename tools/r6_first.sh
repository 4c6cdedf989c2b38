#!/bin/bash
# round 6, first measurement of the three-part bf16 DTW: the f64 three-way test, then the C3 step in every arithmetic
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_first; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dtw_f64.py -x -q -m gpu -s > $O/f64.log 2>&1; echo "f64 rc=$?" >> $O/f64.log
tail -15 $O/f64.log
for rep in 1 2; do
  for v in "f32_matrix 12" "f32_matrix 8" "fast_split 12" "strict_f32 12"; do
    set -- $v
    RP_MFMA3_WAVES=$2 timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 --arith $1 2> $O/b_$1_$2_$rep.err | grep '^{' | tail -1 > $O/b_$1_$2_$rep.json
    python - <<PY
import json
try:
    j=json.loads(open("$O/b_$1_$2_$rep.json").read())
    print("$1 waves $2 rep $rep: %.1f M  step %.3f ms  kernels %s  dtype %s" % (j["value"]/1e6, j["ms_per_step"], j["roofline"]["kernels_ms"], j["dtype"]))
except Exception as e:
    print("$1 $2 rep $rep FAILED", e); print(open("$O/b_$1_$2_$rep.err").read()[-1500:])
PY
  done
done
