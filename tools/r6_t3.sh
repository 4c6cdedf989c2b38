#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_t3; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -300 > $O/gpu_tests.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | head -80
for p in f32 f32_fast f32_strict bf16; do
  timeout 600 python bench.py --mode mlp --mlp-precision $p --no-cpu-baseline 2> $O/c5_$p.err | grep '^{' | tail -1 > $O/c5_$p.json
  python - <<PY
import json
try:
    j=json.loads(open("$O/c5_$p.json").read())
    print("C5 $p: %.1f M rows/s  %.4f ms  frac %.3f  kernel %s" % (j["value"]/1e6, j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("kernel")))
except Exception as e:
    print("C5 $p FAILED", e); print(open("$O/c5_$p.err").read()[-1500:])
PY
done
timeout 600 python bench.py --mode model --no-cpu-baseline 2> $O/model.err | grep '^{' | tail -1 > $O/model.json; python -c "
import json; j=json.loads(open('$O/model.json').read()); print('model detector: %.1f M  %.3f ms' % (j['value']/1e6, j['ms_per_step']), j.get('kernels_ms') or j['config'].get('kernels_ms'))"
