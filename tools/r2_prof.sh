#!/bin/bash
# rocprofv3 kernel-trace + stats of bench.py variants.  Usage: tools/r2_prof.sh <tag> [bench args...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o bench -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_${TAG}_bench.log 2>&1
grep '^{' gpurun_out/prof_${TAG}_bench.log | cut -c1-300
find gpurun_out/prof_${TAG} -name '*kernel_stats.csv' | head -1 | xargs -r cat | cut -c1-200 | head -14
