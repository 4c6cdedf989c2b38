#!/bin/bash
# Runs ON THE GPU BOX: full GPU suite + headline bench + C2 variants + C5 variants.  Usage: tools/r3_full.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r3full}
O=gpurun_out/$TAG; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -15 $O/tests.log
run() { name=$1; shift; timeout 900 python bench.py --steps 10 --warmup 3 "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    j=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    r=j.get("roofline") or {}
    print("$name: %.1f M %s  ms/step %.4f  %s frac %s" % (j["value"]/1e6, j["unit"], j["ms_per_step"], r.get("kernels_ms", j.get("kernels_ms","")), r.get("frac")))
except Exception as e:
    print("$name: FAILED", e); print(open("$O/$name.err").read()[-1500:])
PY
}
run default --no-cpu-baseline
run c2 --no-cpu-baseline --config C2 --steps 50
RP_DTW_NO_SPLIT=1 run c2_nosplit --no-cpu-baseline --config C2 --steps 50
run s4096 --no-cpu-baseline --streams 4096 --steps 20
RP_DTW_NO_SPLIT=1 run s4096_nosplit --no-cpu-baseline --streams 4096 --steps 20
run c5_bf16 --no-cpu-baseline --config C5 --steps 50
RP_MLP_STREAM_WAVES=4 run c5_bf16_w4 --no-cpu-baseline --config C5 --steps 50
run c5_f32 --no-cpu-baseline --config C5 --mlp-precision f32 --steps 50
RP_MLP_STREAM_WAVES=4 run c5_f32_w4 --no-cpu-baseline --config C5 --mlp-precision f32 --steps 50
RP_MLP_STREAM=0 run c5_f32_old --no-cpu-baseline --config C5 --mlp-precision f32 --steps 50
