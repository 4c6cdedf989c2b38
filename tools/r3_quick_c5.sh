#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "mlp or model or c5" > $O/tests.log 2>&1; tail -3 $O/tests.log
for p in bf16 f32; do
  timeout 300 python3 bench.py --config C5 --mlp-precision $p --steps 50 --warmup 5 --no-cpu-baseline > $O/c5_$p.json 2> $O/c5_$p.err
  python3 - <<PY
import json
j=json.loads(open("$O/c5_$p.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("$p: %.1f M rows/s %.4f ms  frac %.3f" % (j["value"]/1e6, r["avg_launch_ms"], r["frac"]))
PY
done
