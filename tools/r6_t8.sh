#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_t8; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -200 > $O/gpu_tests.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | head -40
for m in small medium large; do
 timeout 600 python bench.py --mode model --model-type $m --streams 32768 --steps 10 --warmup 3 --no-cpu-baseline 2> $O/model_$m.err | grep '^{' | tail -1 > $O/model_$m.json; python -c "
import json; j=json.loads(open('$O/model_$m.json').read()); print('model detector $m: %.1f M  %.3f ms' % (j['value']/1e6, j['ms_per_step']), j.get('kernels_ms') or j['config'].get('kernels_ms'), j['config'].get('forward_kernel'))"
done
