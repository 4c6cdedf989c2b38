#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_t2; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -200 > $O/gpu_tests.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_tests.log | head -80
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 1500 $O/bench_default.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r6_t2/bench_default.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["dtype"])
print({k:v for k,v in j["roofline"].items() if not isinstance(v,(dict,list)) and k!="note"})
print({k:v for k,v in j["config"].items() if not isinstance(v,(dict,list))})
print(j.get("cpu_baseline"))
PY
