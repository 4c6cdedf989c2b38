#!/bin/bash
# Runs ON THE GPU BOX (or here, build only): A/B of build variants on ONE box, interleaved (the pool's boxes differ by more than
# most steps).  Every variant is built into its OWN library (rustpotter_amd/variants/lib_<n>.so) from its OWN object directory and
# selected per run with RP_LIB_PATH -- the product library rustpotter_amd/librustpotter_hip.so and csrc/_obj are never touched, so
# no exit path can leave a variant installed (round-3 advice: the old r3_*_ab.sh scripts copied variants over the product).
#
#   tools/ab.sh [-w "<bench args>"]... [-x "<command>"]... [-r reps] -- "<extra flags A>" "<extra flags B>" ...
#   (-x: any command instead of a bench line, run with RP_LIB_PATH set to the variant; its output is printed)
#   e.g. tools/ab.sh -w "--steps 10 --warmup 3" -w "--config C2 --steps 50" -- "" "-DRP_MFMA_GX_PD=1"
# Prints one line per (variant, workload, repetition); the JSON lines stay under gpurun_out/ab/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ab; mkdir -p $O rustpotter_amd/variants
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-pass-failed"
WORK=(); CMDS=(); REPS=2
while [ $# -gt 0 ]; do
  case "$1" in
    -w) WORK+=("$2"); shift 2;;
    -x) CMDS+=("$2"); shift 2;;
    -r) REPS=$2; shift 2;;
    --) shift; break;;
    *) break;;
  esac
done
[ ${#WORK[@]} -eq 0 ] && [ ${#CMDS[@]} -eq 0 ] && WORK=("--steps 10 --warmup 3")
n=0
for flags in "$@"; do
  make -C rustpotter_amd/csrc -j8 OUT=../variants/lib_$n.so OBJDIR=_obj_v$n CXXFLAGS="$BASE $flags" > $O/make_$n.log 2>&1 || { tail -5 $O/make_$n.log; exit 1; }
  n=$((n+1))
done
[ -n "$AB_BUILD_ONLY" ] && exit 0
for rep in $(seq 1 $REPS); do
  i=0
  for flags in "$@"; do
    wi=0
    for w in "${WORK[@]}"; do
      RP_LIB_PATH=$PWD/rustpotter_amd/variants/lib_$i.so timeout 900 python3 bench.py --no-cpu-baseline --no-extras $w 2> $O/v${i}_w${wi}_$rep.err | grep '^{' | tail -1 > $O/v${i}_w${wi}_$rep.json
      python3 - <<PY
import json
try:
    j=json.loads(open("$O/v${i}_w${wi}_$rep.json").read())
    k=(j.get("roofline") or {}).get("kernels_ms") or j.get("kernels_ms") or j["config"].get("kernels_ms")
    print("variant $i [$flags] rep $rep [$w]: %.1f M %s  step %.4f ms  kernels %s  build %s" % (j["value"]/1e6, j["unit"], j["ms_per_step"], k, (j.get("build") or {}).get("flags_extra")))
except Exception as e:
    print("variant $i [$flags] rep $rep [$w]: FAILED", e); print(open("$O/v${i}_w${wi}_$rep.err").read()[-800:])
PY
      wi=$((wi+1))
    done
    for c in "${CMDS[@]}"; do
      [ $rep -eq 1 ] && { echo "variant $i [$flags] \$ $c"; RP_LIB_PATH=$PWD/rustpotter_amd/variants/lib_$i.so timeout 900 bash -c "$c" 2>&1 | grep -v amdgpu.ids; }
    done
    i=$((i+1))
  done
done
rm -rf rustpotter_amd/variants rustpotter_amd/csrc/_obj_v*
