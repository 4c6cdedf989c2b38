#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3misc; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3.py -q -x -k "generic" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "api" > $O/tests_api.log 2>&1; tail -3 $O/tests_api.log
for sig in noise silence hum; do for a in 0.0 0.2; do echo "latency SIGNAL=$sig AVG=$a: $(SIGNAL=$sig AVG=$a python tools/latency_probe.py 2>/dev/null | tail -1)"; done; done | tee $O/latency.txt
