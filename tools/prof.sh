#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): every profiling / evidence recipe of this repository in one place.  Output goes under gpurun_out/
# (scratch); the collect_* scripts copy what is kept into profiles/.  <cmd...> is a python command line WITHOUT the interpreter
# (rocprofv3 must start the program itself: no env / bash -c hops), e.g. `bench.py --config C2` or `tools/bench_frontend.py 65536 1 1 i16`.
#
#   prof.sh stats  <tag> <cmd...>              rocprofv3 --kernel-trace --stats; prints the per-kernel medians   -> gpurun_out/prof_<tag>/
#   prof.sh pmc    <tag> "<C1 C2 ..>" <cmd...> ONE --pmc pass (kernel trace only beside it)                       -> gpurun_out/pmc_<tag>/
#   prof.sh pmcset <tag> <cmd...>              the standard passes, one run each: <tag>_fetch _write _sq _inst _valu _clk _lds
#   prof.sh trace  <tag> <cmd...>              kernel trace: average duration per kernel and the gaps between consecutive kernels
#   prof.sh c5     <bf16|f32>                  the C5 evidence of one precision (bench line + stats + PMC)        -> gpurun_out/c5_<precision>/
#   prof.sh round                              everything profiles/ keeps for a round: stats + pmcset of the headline command (tags final*),
#                                              c5 bf16 / f32, tools/run_final_benches.sh.  Then, in the container:
#                                              python tools/collect_profiles.py rNN; python tools/collect_c5.py rNN c5_bf16 bf16; ... c5_f32 f32
#   prof.sh rounds [sizes...]                  DTW time against the number of tile rounds around BASELINE config C2
#   prof.sh sweeps [seed]                      the randomised parity sweeps beyond the test suite (tests/sweep_parity.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
SUB=${1:?subcommand}; shift
medians() {   # per-kernel call count / median / min of a kernel trace
python3 - "$1" <<'PY'
import csv, sys, glob, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not f:
    raise SystemExit("no kernel trace under " + sys.argv[1])
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "rp::" in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("   %-84s calls %4d  median %9.4f ms  min %9.4f  total %9.3f" % (k[:84], len(v), statistics.median(v), min(v), sum(v)))
PY
}
counters() {
python3 - "$1" <<'PY'
import csv, sys, glob, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    raise SystemExit("no counter csv under " + sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: "%.5g" % statistics.median(v) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
}
case "$SUB" in
stats)
    TAG=$1; shift
    rm -rf gpurun_out/prof_$TAG
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python3 "$@" > gpurun_out/prof_${TAG}.log 2>&1
    grep '^{' gpurun_out/prof_${TAG}.log | cut -c1-400
    medians gpurun_out/prof_$TAG ;;
pmc)
    TAG=$1; CTRS=$2; shift 2
    rm -rf gpurun_out/pmc_$TAG
    rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/pmc_$TAG -o pmc -- python3 "$@" > gpurun_out/pmc_${TAG}.log 2>&1
    counters gpurun_out/pmc_$TAG ;;
pmcset)
    TAG=$1; shift
    bash tools/prof.sh pmc ${TAG}_fetch "FETCH_SIZE" "$@" > /dev/null 2>&1
    bash tools/prof.sh pmc ${TAG}_write "WRITE_SIZE" "$@" > /dev/null 2>&1
    bash tools/prof.sh pmc ${TAG}_sq "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "$@"
    bash tools/prof.sh pmc ${TAG}_inst "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "$@" > /dev/null 2>&1
    bash tools/prof.sh pmc ${TAG}_valu "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" "$@"
    bash tools/prof.sh pmc ${TAG}_clk "GRBM_GUI_ACTIVE" "$@" > /dev/null 2>&1
    bash tools/prof.sh pmc ${TAG}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "$@" ;;
trace)
    TAG=$1; shift
    O=gpurun_out/trace_$TAG; rm -rf $O; mkdir -p $O
    rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 "$@" > $O/log.txt 2>&1
    python3 - "$O" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "rp::" in r["Kernel_Name"]]
short = lambda n: n.split("(")[0].replace("void ", "").replace("rp::", "")[:60]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[short(a["Kernel_Name"])].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap[short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k, v in dur.items(): print("%-62s n=%4d  avg %8.2f us  min %8.2f" % (k, len(v), sum(v[len(v)//2:]) / len(v[len(v)//2:]) / 1e3, min(v) / 1e3))
for k, v in gap.items():
    if len(v) > 10: print("gap %-110s avg %7.2f us" % (k, sum(v[len(v)//2:]) / len(v[len(v)//2:]) / 1e3))
PY
    ;;
c5)
    P=${1:-bf16}
    O=gpurun_out/c5_$P; mkdir -p $O
    ARGS="bench.py --config C5 --mlp-precision $P --no-cpu-baseline"
    timeout 600 python3 $ARGS --steps 50 --warmup 5 > $O/c5_$P.json 2> $O/c5_$P.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 $ARGS --steps 50 --warmup 5 > $O/prof_c5.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
        tag=$(echo $c | cut -d' ' -f1)
        rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -o pmc -- python3 $ARGS --steps 10 --warmup 2 > $O/pmc_$tag.log 2>&1
    done
    cut -c1-300 $O/c5_$P.json ;;
round)
    HEAD="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
    bash tools/prof.sh stats final $HEAD > gpurun_out/final_prof.txt 2>&1
    bash tools/prof.sh pmcset final bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/final_pmc.txt 2>&1
    bash tools/prof.sh c5 bf16 > gpurun_out/final_c5_bf16.txt 2>&1
    bash tools/prof.sh c5 f32 > gpurun_out/final_c5_f32.txt 2>&1
    bash tools/run_final_benches.sh > gpurun_out/final_benches.txt 2>&1
    tail -45 gpurun_out/final_benches.txt ;;
rounds)
    for s in ${@:-166 331 662 993 1024 1324 2048 4096}; do
        python3 bench.py --streams $s --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['roofline']['kernels_ms']
tiles=$s*297/32.0
print('streams %5d  tiles %7.0f  rounds %.2f  step %.4f ms  mfcc %.4f  dtw %.4f  scan %.4f  M/s %.1f  dtw us per round %.1f' % ($s, tiles, tiles/3072, j['ms_per_step'], k['mfcc'], k['dtw'], k['scan'], j['value']/1e6, k['dtw']*1e3/(tiles/3072)))"
    done ;;
sweeps)
    SEED=${1:-17}
    mkdir -p gpurun_out/sweep
    timeout 5000 python3 tests/sweep_parity.py --seed $SEED --cases 600 --mfma-cases 1500 --ragged-cases 700 --api-cases 200 --live-multi-cases 400 --multi-cases 200 --model-cases 200 \
        --reset-cases 150 --rate-cases 80 --mfcc-cases 1500 --frontend-cases 60 --resample-cases 40 --builder-cases 30 --train-cases 10 --extreme-cases 150 2>&1 |
        grep -v "case [0-9]* ok\|amdgpu.ids" > gpurun_out/sweep/sweep_$SEED.txt
    tail -25 gpurun_out/sweep/sweep_$SEED.txt ;;
*)
    echo "unknown subcommand $SUB" >&2; exit 2 ;;
esac
