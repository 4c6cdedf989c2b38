#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace + stats of the default
# bench.py command; CSVs land in gpurun_out/prof_<tag>/ and are then copied into
# profiles/ by hand.  Usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o bench -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/prof_${TAG}_bench.log 2>&1
grep '^{' gpurun_out/prof_${TAG}_bench.log | cut -c1-400
find gpurun_out/prof_${TAG} -name '*kernel_stats.csv' | head -1 | xargs -r cat | cut -c1-220 | head -12
