import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import sweep_parity as sp, rpw_py
import rustpotter_amd as ra
from oracle import rp_oracle as orc
ctx = ra.BatchContext(0)
case = sp.make_model_case(np.random.default_rng([202, 99, 874]))
c, m, x = case["cfg"], case["model"], case["x"]
d = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"], vad_mode=c["vad_mode"])
d.add_model(m)
refs = []
for k in range(len(x) // 480):
    r = d.process_f32(x[480 * k:480 * (k + 1)])
    if r is not None: refs.append((k, r["name"], r["counter"], float(r["score"]), float(r["avg_score"]), {a: float(b) for a, b in r["scores"].items()}))
dc = ra.DetectorConfig()
dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
nl = len([k for k in m["weights"] if k.endswith(".weight")])
model = ra.Model(ctx, [m["weights"]["ln%d.weight" % (i + 1)] for i in range(nl)], [m["weights"]["ln%d.bias" % (i + 1)] for i in range(nl)])
det, dlab, n_det = ctx.batch_detect_model(x[None, :], model, m["mfcc_size"], m["labels"].index("none") if "none" in m["labels"] else -1, dc, max_det=64)
print("oracle ", refs)
print("batched", [(int(det[0][j]["frame"]) // 3 + 1, m["labels"][dlab[0][j]], int(det[0][j]["counter"]), float(det[0][j]["score"]), float(det[0][j]["avg_score"])) for j in range(n_det[0])])
d2 = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"], vad_mode=c["vad_mode"])
d2.add_model(m)
rc = ra.RustpotterConfig.default(); rc.fmt.sample_format = ra.SampleFormat.F32
rc.detector.avg_threshold, rc.detector.threshold, rc.detector.min_scores, rc.detector.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
rp = ra.Rustpotter.new(rc)
rp.add_wakeword_from_buffer("m", rpw_py.dump_rpw_model(m["labels"], m["train_size"], m["mfcc_size"], m["m_type"], m["weights"], m["rms_level"]))
for k in range(len(x) // 480):
    r = d2.process_f32(x[480 * k:480 * (k + 1)]); g = rp.process_samples(np.ascontiguousarray(x[480 * k:480 * (k + 1)]))
    st = d2.state(); p = rp.get_partial_detection()
    if 44 <= k <= 57: print(k, "oracle partial", st["partial_counter"], round(float(st["partial_score"]), 6), "countdown", st["countdown"], "| api partial", None if p is None else (p.counter, round(float(p.score), 6)), "| emitted", None if r is None else round(float(r["score"]), 6))
# batched on prefixes
for k in range(50, 57):
    det, dlab, n_det = ctx.batch_detect_model(x[None, :480 * (k + 1)], model, m["mfcc_size"], 0, dc, max_det=64)
    print("prefix", k, [(int(det[0][j]["frame"]), int(det[0][j]["window"]), int(det[0][j]["counter"]), round(float(det[0][j]["score"]), 6)) for j in range(n_det[0])])
