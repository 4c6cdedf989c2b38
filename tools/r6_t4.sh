#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_t4; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -300 > $O/gpu_tests.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | head -60
c5() { # name, precision, env...
  n=$1; p=$2; shift 2
  env "$@" timeout 600 python bench.py --mode mlp --mlp-precision $p --no-cpu-baseline 2> $O/c5_$n.err | grep '^{' | tail -1 > $O/c5_$n.json
  python - <<PY
import json
try:
    j=json.loads(open("$O/c5_$n.json").read())
    print("C5 $n: %.1f M rows/s  %.4f ms  frac %.3f  kernel %s" % (j["value"]/1e6, j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("kernel")))
except Exception as e:
    print("C5 $n FAILED", e); print(open("$O/c5_$n.err").read()[-1500:])
PY
}
for rep in 1 2; do
c5 f32_6waves_d2 f32 A=1
c5 f32_8waves_d1 f32 RP_MLP_STREAM_WAVES=8
c5 fast_d2 f32_fast A=1
c5 fast_d1 f32_fast RP_MLP_STREAM_DEPTH=1
done
for pad in 0 14000 41000 0 14000 41000; do
  RP_MFCC_LDS_PAD=$pad timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 2> $O/mfcc_pad.err | grep '^{' | tail -1 > $O/mfcc_pad_$pad.json
  python -c "
import json; j=json.loads(open('$O/mfcc_pad_$pad.json').read()); print('mfcc lds pad $pad: %.1f M  step %.3f ms  kernels %s' % (j['value']/1e6, j['ms_per_step'], j['roofline']['kernels_ms']))"
done
