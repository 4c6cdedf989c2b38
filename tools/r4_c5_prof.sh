#!/bin/bash
# Runs ON THE GPU BOX: the C5 evidence for one precision (bf16 | f32): bench line, kernel-trace stats, PMC passes.
# Usage: tools/r4_c5_prof.sh bf16|f32     -> gpurun_out/r4c5_<precision>/ (tools/collect_c5.py r04 r4c5_<precision> <precision>)
P=${1:-bf16}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4c5_$P; mkdir -p $O
ARGS="--config C5 --mlp-precision $P"
timeout 600 python3 bench.py $ARGS --steps 50 --warmup 5 > $O/c5_$P.json 2> $O/c5_$P.err
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 bench.py $ARGS --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_c5.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -o pmc -- python3 bench.py $ARGS --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_$tag.log 2>&1
done
cut -c1-300 $O/c5_$P.json
