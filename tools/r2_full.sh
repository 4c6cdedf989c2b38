#!/bin/bash
# full GPU suite + headline bench + the unfriendly shapes.  Usage: tools/r2_full.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1
mkdir -p gpurun_out/$TAG
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/$TAG/tests.log
tail -12 gpurun_out/$TAG/tests.log
run() { name=$1; shift; timeout 900 python bench.py --steps 10 --warmup 3 "$@" > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err; python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/$TAG/$name.json").read().strip().splitlines()[-1])
    print("$name: %.1f M %s  ms/step %.3f  %s %s" % (j["value"]/1e6, j["unit"], j["ms_per_step"], j.get("roofline",{}).get("kernels_ms",""), {k:v for k,v in j["config"].items() if k.startswith(("gate","avg_"))}))
except Exception as e:
    print("$name: FAILED", e); print(open("gpurun_out/$TAG/$name.err").read()[-1500:])
PY
}
run default
run ragged5 --no-cpu-baseline --template-lens 108,96,90,93,102
run t3 --no-cpu-baseline --templates 3 --template-len 126
run median --no-cpu-baseline --score-mode median
run gate02 --no-cpu-baseline --avg-gate
run gate04 --no-cpu-baseline --avg-gate --avg-threshold 0.4
run gate04_full --no-cpu-baseline --avg-gate --avg-threshold 0.4 --full-scores
run c2 --no-cpu-baseline --streams 1024
run c4share --no-cpu-baseline --streams 8192 --templates 64
