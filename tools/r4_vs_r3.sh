cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { # dir args...
  d=$1; shift
  (cd $d && python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); k=(j.get('roofline') or {}).get('kernels_ms') or j['config'].get('kernels_ms')
print('%-4s %-60s %8.1f M  step %.4f ms  %s' % ('$d', '$*', j['value']/1e6, j['ms_per_step'], k))")
}
for rep in 1 2; do
 for d in _r3 .; do
  X=""; [ "$d" = "." ] && X="--no-extras"
  run $d --streams 8192 --templates 64 --steps 10 --warmup 3 $X
  run $d --config C2 --steps 50 --warmup 5 $X
  run $d --steps 10 --warmup 3 $X
  run $d --mode stream --chunks-per-call 1 --steps 20 --warmup 5 $X
 done
done
