#!/bin/bash
# The 1/2/4/8-GPU scaling curve of the MFCC+DTW path on ONE node, one rank per GPU over RCCL:
#   weak scaling   -- C3 per GPU (65 536 streams x 8 templates each), value = scorings/s of the whole job
#   strong scaling -- BASELINE config C4 (65 536 streams x 64 templates SPLIT over the ranks by stream)
# Every line is a plain `bench.py --gpus N` line (self-proving: rccl_world_size, per-rank device UUIDs, gather_ms_per_step with the
# bytes of the per-stream result block); the last line of the output file is the efficiency table computed from them.
# usage: tools/run_scale.sh [out.jsonl]   (N runs up to the number of visible devices; needs no arguments on an 8-GPU box)
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale.jsonl}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
NDEV=$(python3 -c 'import torch; print(torch.cuda.device_count())')
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29511
for N in 1 2 4 8; do
    [ "$N" -le "$NDEV" ] || { echo "{\"skipped\": \"--gpus $N: only $NDEV device(s) visible\"}" >> "$OUT"; continue; }
    for MODE in weak strong; do
        ARGS="--gpus $N --steps 10 --warmup 3 --no-extras --no-cpu-baseline"
        [ "$MODE" = strong ] && ARGS="$ARGS --config C4"
        if [ "$N" -eq 1 ]; then
            python3 bench.py $ARGS >> "$OUT" 2>> "${OUT%.jsonl}.err"
        else
            PORT=$((PORT + 1))
            python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" bench.py $ARGS >> "$OUT" 2>> "${OUT%.jsonl}.err"
        fi
    done
done
python3 - "$OUT" <<'EOF'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
curve = {"weak": {}, "strong": {}}
for r in rows:
    if "value" in r:
        curve[r["scaling"]][r["n_gpus"]] = r
tab = {}
for mode, by_n in curve.items():
    if 1 not in by_n:
        continue
    base = by_n[1]["value"]
    tab[mode] = {str(n): {"value": r["value"], "ms_per_step": r["ms_per_step"], "speedup": r["value"] / base, "efficiency": r["value"] / base / n,
                          "gather_ms_per_step": (r["config"].get("gather_ms_per_step") or {}).get("mean"),
                          "rccl_world_size": r["config"].get("rccl_world_size"), "distinct_devices": r["config"].get("distinct_devices")}
                 for n, r in sorted(by_n.items())}
line = {"scaling_table": tab, "unit": "scorings/s", "note": "efficiency = value(N) / (N x value(1)); weak: 65 536 streams x 8 templates per GPU, "
        "strong: 65 536 streams x 64 templates split over the ranks"}
open(sys.argv[1], "a").write(json.dumps(line) + "\n")
print(json.dumps(line, indent=1))
EOF
