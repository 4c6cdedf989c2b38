#!/bin/bash
# Runs ON THE GPU BOX: builds tools/scratch/${PROBE:-dtw_mfma_probe}.hip in the shapes given as arguments ("NWAVES WGS_PER_CU [flags]"), checks
# each against the CPU and times it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3dtwmfma; mkdir -p $O
[ $# -eq 0 ] && set -- "8 1 -DRP_RAW_RSQ" "8 1" "4 2 -DRP_RAW_RSQ"
for cfg in "$@"; do
  read -r nw wg flags <<< "$cfg"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DNWAVES=$nw -DWGS_PER_CU=$wg $flags tools/scratch/${PROBE:-dtw_mfma_probe}.hip -o /tmp/dtw_mfma_probe_x 2> $O/build.log || { echo "build failed $cfg"; tail -5 $O/build.log; continue; }
  echo "== NWAVES=$nw WGS_PER_CU=$wg $flags"
  timeout 120 /tmp/dtw_mfma_probe_x ${ARGS:-100 8192 288} 2>&1 | tail -${TAILN:-5}
done
