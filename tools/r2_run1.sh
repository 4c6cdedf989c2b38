#!/bin/bash
# round-2 first GPU pass: new tests, default bench, avg-gate bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests/test_gpu_round2.py tests/test_rpw_fuzz.py -m gpu -q -x > gpurun_out/r2a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2a/tests.log
tail -25 gpurun_out/r2a/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err
timeout 600 python bench.py --steps 10 --warmup 3 --avg-gate --no-cpu-baseline > gpurun_out/r2a/bench_gate.json 2> gpurun_out/r2a/bench_gate.err
timeout 600 python bench.py --steps 10 --warmup 3 --avg-gate --full-scores --no-cpu-baseline > gpurun_out/r2a/bench_gate_full.json 2> gpurun_out/r2a/bench_gate_full.err
for f in bench_default bench_gate bench_gate_full; do echo "== $f"; cut -c1-1500 gpurun_out/r2a/$f.json; tail -3 gpurun_out/r2a/$f.err; done
