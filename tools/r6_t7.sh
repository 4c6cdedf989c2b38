#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_t7; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_model_pin.py tests/test_gpu_rccl.py -m gpu -q 2>&1 | tail -3
for rep in 1 2 3; do
  for w in 8 12; do
    RP_MFMA3_WAVES=$w timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2> $O/b_$w.err | grep '^{' | tail -1 > $O/b_${w}_$rep.json
    python -c "
import json; j=json.loads(open('$O/b_${w}_$rep.json').read()); print('waves $w rep $rep: %.1f M  step %.3f ms  kernels %s' % (j['value']/1e6, j['ms_per_step'], j['roofline']['kernels_ms']))"
  done
done
for w in 8 12; do
RP_MFMA3_WAVES=$w timeout 600 python bench.py --no-extras --no-cpu-baseline --config C2 --steps 50 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('C2 waves $w: %.1f M  step %.4f ms  kernels %s' % (j['value']/1e6, j['ms_per_step'], j['roofline']['kernels_ms']))"
RP_MFMA3_WAVES=$w timeout 600 python bench.py --no-extras --no-cpu-baseline --streams 8192 --templates 64 --steps 10 --warmup 3 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('C4 share waves $w: %.1f M  step %.4f ms  kernels %s' % (j['value']/1e6, j['ms_per_step'], j['roofline']['kernels_ms']))"
done
