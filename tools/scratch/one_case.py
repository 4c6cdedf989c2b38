import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import sweep_parity as sp
import rustpotter_amd as ra
ctx = ra.BatchContext(0)
seed, ci = int(sys.argv[1]), int(sys.argv[2])
case = sp.make_case(np.random.default_rng([seed, ci]), extreme=len(sys.argv) > 3)
ref = sp.oracle_detections(case)
off, live, agg = sp.device_detections(ra, ctx, case)
print(case["cfg"], case["K"], [len(t) for t in case["templates"]], case["pcm"].shape, case["pcm"].dtype, case["chunks_per_call"])
print("oracle ", ref); print("offline", off); print("live   ", live)
thr = case["cfg"]["threshold"]
print("closest agg to threshold:", float(np.min(np.abs(agg - np.float32(thr)))), "agg range", float(agg.min()), float(agg.max()))
