#!/bin/bash
# A/B of experiment builds of the library on one box, interleaved.  Usage: tools/scratch/ab.sh <rounds> <name> [<name> ...]   (BASE = the in-tree library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$1; shift
for r in $(seq $R); do for a in "$@"; do
  if [ $a = BASE ]; then unset RP_LIB_PATH; else export RP_LIB_PATH=$PWD/tools/scratch/exp_lib_$a.so; fi
  echo -n "$a "; python bench.py --steps 8 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['roofline']['kernels_ms'])"
done; done
