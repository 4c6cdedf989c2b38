#!/bin/bash
# Runs ON THE GPU BOX: streaming-read probe + C5 baseline (bench line, kernel-trace stats, FETCH/WRITE PMC passes).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3p1
( cd tools/scratch && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o glds_probe glds_probe.hip && timeout 300 ./glds_probe ) > gpurun_out/r3p1/glds_probe.txt 2>&1
cat gpurun_out/r3p1/glds_probe.txt
for p in bf16 f32; do
  timeout 600 python3 bench.py --mode mlp --mlp-precision $p --steps 50 --warmup 5 > gpurun_out/r3p1/c5_$p.json 2> gpurun_out/r3p1/c5_$p.err
  tail -c 600 gpurun_out/r3p1/c5_$p.json
done
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3p1/prof_c5 -o c5 -- python3 bench.py --mode mlp --steps 20 --warmup 3 > gpurun_out/r3p1/prof_c5.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r3p1/pmc_c5_$c -o pmc -- python3 bench.py --mode mlp --steps 5 --warmup 2 > gpurun_out/r3p1/pmc_c5_$c.log 2>&1
done
find gpurun_out/r3p1 -name "*kernel_stats.csv" | head -3 | xargs -r head -8
