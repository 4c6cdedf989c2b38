#!/bin/bash
# quick GPU check: selected parity tests + the default bench line.  Usage: tools/r2_quick.sh <tag> "<pytest -k expr>" [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; KEXPR=$2; shift 2
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -q -x -k "$KEXPR" > gpurun_out/$TAG/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/$TAG/tests.log
tail -8 gpurun_out/$TAG/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print("value %.1f M/s  ms/step %.3f  kernels %s" % (j["value"]/1e6, j["ms_per_step"], j["roofline"]["kernels_ms"]))
PY
tail -2 gpurun_out/$TAG/bench.err
