#!/bin/bash
# Runs ON THE GPU BOX: MLP parity tests on the stream kernel, C5 bench old vs new, probe variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3c5b
timeout 900 python -m pytest tests -m gpu -q -x -k "mlp or model or c5" > gpurun_out/r3c5b/tests.log 2>&1; tail -5 gpurun_out/r3c5b/tests.log
for p in bf16 f32; do
  for sm in 1 0; do
    RP_MLP_STREAM=$sm timeout 600 python3 bench.py --mode mlp --mlp-precision $p --steps 50 --warmup 5 > gpurun_out/r3c5b/c5_${p}_stream$sm.json 2> gpurun_out/r3c5b/c5_${p}_stream$sm.err
    echo "$p stream=$sm: $(tail -c 330 gpurun_out/r3c5b/c5_${p}_stream$sm.json)"
  done
done
( cd tools/scratch && timeout 300 ./glds_probe ) > gpurun_out/r3c5b/glds_probe.txt 2>&1
cat gpurun_out/r3c5b/glds_probe.txt
