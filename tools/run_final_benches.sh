#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): every bench line that profiles/ keeps, into gpurun_out/final/*.json
# (tools/collect_profiles.py then copies them into profiles/).  Usage: tools/run_final_benches.sh [all]
# Without `all` only the lines a round usually moves are run (headline, C2, C4 share, the ragged shapes in both forms, mfcc_size 16, C5);
# the rest of the catalogue (gates, live streams, resampler, model detector, ingest, latency) is re-measured when its code changed.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/final
O=gpurun_out/final
B="python3 bench.py --steps 10 --warmup 3 --no-extras"
line() { grep '^{' | tail -1; }
python3 bench.py --steps 10 --warmup 3 2>/dev/null | line > $O/bench_default.json
$B --no-cpu-baseline --config C2 --steps 50 2>/dev/null | line > $O/c2.json
$B --no-cpu-baseline --streams 8192 --templates 64 2>/dev/null | line > $O/c4.json
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --config C4 2>/dev/null | line > $O/c4_one_gpu.json
RP_BENCH_OVERSUBSCRIBE=1 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --config C4 --gpus 2 2>/dev/null | line > $O/c4_two_ranks_one_gpu.json
# the same shapes in the opt-in two-part f16 arithmetic (RP_ARITH_FAST_SPLIT: 22-bit products, narrower than the reference) and in strict f32
$B --no-cpu-baseline --arith fast_split 2>/dev/null | line > $O/c3_fast_split.json
$B --no-cpu-baseline --arith strict_f32 2>/dev/null | line > $O/c3_strict_f32.json
$B --no-cpu-baseline --arith fast_split --config C2 --steps 50 2>/dev/null | line > $O/c2_fast_split.json
$B --no-cpu-baseline --arith fast_split --streams 8192 --templates 64 2>/dev/null | line > $O/c4_fast_split.json
$B --no-cpu-baseline --template-lens 108,96,90,93,102 2>/dev/null | line > $O/ragged5.json
$B --no-cpu-baseline --arith fast_split --ragged-matrix --template-lens 108,96,90,93,102 2>/dev/null | line > $O/ragged5_matrix.json
$B --no-cpu-baseline --template-lens 117,126,99 2>/dev/null | line > $O/ragged3_alexa.json
$B --no-cpu-baseline --arith fast_split --ragged-matrix --template-lens 117,126,99 2>/dev/null | line > $O/ragged3_alexa_matrix.json
$B --no-cpu-baseline --streams 8192 --mfcc-size 16 2>/dev/null | line > $O/k16.json
$B --no-cpu-baseline --arith fast_split --streams 8192 --mfcc-size 16 2>/dev/null | line > $O/k16_fast_split.json
$B --no-cpu-baseline --arith strict_f32 --streams 8192 --mfcc-size 16 2>/dev/null | line > $O/k16_strict_f32.json
$B --no-cpu-baseline --streams 8192 --mfcc-size 13 2>/dev/null | line > $O/k13.json
$B --no-cpu-baseline --arith strict_f32 --streams 8192 --mfcc-size 13 2>/dev/null | line > $O/k13_strict_f32.json
$B --no-cpu-baseline --templates 4 2>/dev/null | line > $O/t4.json
$B --no-cpu-baseline --arith strict_f32 --templates 4 2>/dev/null | line > $O/t4_strict_f32.json
$B --no-cpu-baseline --arith fast_split --templates 4 2>/dev/null | line > $O/t4_fast_split.json
$B --no-cpu-baseline --mode mlp --mlp-precision bf16 2>/dev/null | line > $O/c5_bf16.json
$B --no-cpu-baseline --mode mlp --mlp-precision f32 2>/dev/null | line > $O/c5_f32.json
$B --no-cpu-baseline --mode mlp --mlp-precision f32_fast 2>/dev/null | line > $O/c5_f32_fast.json
$B --no-cpu-baseline --mode mlp --mlp-precision f32_strict 2>/dev/null | line > $O/c5_f32_strict.json
$B --no-cpu-baseline --detect-only --template-lens 108,96,90,93,102 2>/dev/null | line > $O/detect_only_ragged5.json
if [ "${1:-}" != all ]; then for f in $O/*.json; do echo "$f $(cut -c1-160 $f)"; done; exit 0; fi
$B --no-cpu-baseline --templates 3 --template-len 126 2>/dev/null | line > $O/t3.json
$B --no-cpu-baseline --score-mode median 2>/dev/null | line > $O/median.json
$B --no-cpu-baseline --detect-only 2>/dev/null | line > $O/detect_only.json
$B --no-cpu-baseline --avg-gate 2>/dev/null | line > $O/gate_default.json
$B --no-cpu-baseline --avg-gate --full-scores 2>/dev/null | line > $O/gate_default_full.json
$B --no-cpu-baseline --avg-gate --avg-threshold 0.4 2>/dev/null | line > $O/gate_04.json
$B --no-cpu-baseline --avg-gate --avg-threshold 0.4 --full-scores 2>/dev/null | line > $O/gate_04_full.json
RP_BENCH_OVERSUBSCRIBE=1 $B --no-cpu-baseline --gpus 2 --streams 32768 2>/dev/null | line > $O/two_ranks_one_gpu.json
$B --no-cpu-baseline --mode stream --chunks-per-call 1 2>/dev/null | line > $O/stream1.json
$B --no-cpu-baseline --mode stream --chunks-per-call 8 2>/dev/null | line > $O/stream8.json
$B --no-cpu-baseline --mode resample --streams 8192 2>/dev/null | line > $O/rs_fft.json
RP_RESAMPLE_GEMM=1 $B --no-cpu-baseline --mode resample --streams 8192 2>/dev/null | line > $O/rs_gemm.json
$B --no-cpu-baseline --mode resample --streams 8192 --pcm-format i16 --channels 2 2>/dev/null | line > $O/rs_fft_i16_stereo.json
for m in small medium large; do python3 bench.py --mode model --model-type $m --streams 32768 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line > $O/model_$m.json; done
python3 bench.py --ingest --ingest-format f32 --no-cpu-baseline 2>/dev/null | line > $O/ingest_f32.json
python3 bench.py --ingest --ingest-format i16 --no-cpu-baseline 2>/dev/null | line > $O/ingest_i16.json
bash tools/prof.sh rounds > $O/c2_rounds.txt 2>/dev/null
python3 tools/bench_model_detect.py > $O/model_detect.txt 2>/dev/null
for sig in noise silence; do for a in 0.0 0.2 0.5; do echo "single-stream API, $sig, avg_threshold $a: $(SIGNAL=$sig AVG=$a python3 tools/latency_probe.py 2>/dev/null | tail -1)"; done; done > $O/latency.txt
python3 tools/bench_frontend.py > $O/frontend.txt 2>/dev/null
for f in $O/*.json; do echo "$f $(cut -c1-160 $f)"; done
cat $O/*.txt
