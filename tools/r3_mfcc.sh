#!/bin/bash
# Runs ON THE GPU BOX: MFCC parity tests + headline bench (kernel times)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3mfcc; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "mfcc or silence or fixture or golden or stream_batch_equals" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/default_$i.json 2> $O/default_$i.err
python - <<PY
import json
j=json.loads(open("$O/default_$i.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("default: %.1f M  ms/step %.3f  %s" % (j["value"]/1e6, j["ms_per_step"], r["kernels_ms"]))
PY
done
