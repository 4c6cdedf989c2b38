#!/bin/bash
# Runs ON THE GPU BOX: GPU suite (all failures listed) + the default bench with the matrix-core DTW kernel on and off.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3mfmatry; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round3.py::test_live_multi_sweep_few_cases 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" > $O/tests.log
tail -25 $O/tests.log
for m in 1 0; do
  RP_DTW_MFMA=$m timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$m.json 2> $O/bench_$m.err
  python - <<PY
import json
j=json.loads(open("$O/bench_$m.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("RP_DTW_MFMA=$m: %.1f M scorings/s, step %.3f ms, kernels %s" % (j["value"]/1e6, j["ms_per_step"], r.get("kernels_ms")))
PY
done
