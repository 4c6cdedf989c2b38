#!/bin/bash
# Runs ON THE GPU BOX: mfcc tests + interleaved benches: direct-load kernel (default) vs staged (RP_MFCC_STAGED=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3mfccenv; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "mfcc or silence or fixture or golden or stream_batch_equals or smoke or c3" > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2 3; do
  for st in 0 1; do
    RP_MFCC_STAGED=$st timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA > $O/b_${st}_$rep.json 2> $O/b_${st}_$rep.err
    python - <<PY
import json
j=json.loads(open("$O/b_${st}_$rep.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("staged=$st rep $rep: mfcc %.4f dtw %.4f  step %.3f  %.1f M" % (r["kernels_ms"]["mfcc"], r["kernels_ms"]["dtw"], j["ms_per_step"], j["value"]/1e6))
PY
  done
done
