import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import sweep_parity as sp
import rustpotter_amd as ra
ctx = ra.BatchContext(0)
for seed, ci in ((4, 382), (2, 91), (3, 117)):
    case = sp.make_case(np.random.default_rng([seed, ci]))
    c = case["cfg"]
    dc = ra.DetectorConfig()
    dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
    dc.score_ref, dc.band_size = c["score_ref"], c["band_size"]
    dc.score_mode = getattr(ra.ScoreMode, c["score_mode"].capitalize())
    tm = ra.Templates(ctx, case["templates"], avg=case["avg"])
    pcm = case["pcm"]
    det, n_det, scores, agg = ctx.batch_detect(pcm, tm, dc, max_det=32, want_scores=True)
    sb = ra.StreamBatch(ctx, tm, dc, pcm.shape[0], max_chunks_per_call=case["chunks_per_call"])
    step = 480 * case["chunks_per_call"]; n = (pcm.shape[1] // 480) * 480; L = tm.max_len
    nd_total = 0; diffs = []
    for i in range(0, n, step):
        piece = np.ascontiguousarray(pcm[:, i:min(i + step, n)])
        d, nd, a = sb.process(piece, max_det=8, want_agg=True)
        f0 = 3 * (i // 480) - 3
        for k in range(a.shape[1]):
            wi = f0 + k - L + 1
            if 0 <= wi < agg.shape[1] and a[0, k] != agg[0, wi]:
                diffs.append((wi, float(a[0, k]), float(agg[0, wi])))
    print(seed, ci, 'S', pcm.shape, 'cpc', case["chunks_per_call"], 'avg', None if case["avg"] is None else len(case["avg"]), 'L', L, 'n_win', agg.shape[1], 'ndiff', len(diffs), diffs[:5])
    # does the offline result depend on the window range?  score windows one at a time through dtw_scores
    mf = ctx.mfcc(pcm, case["K"])
    sc_all, _, agg_all = ctx.dtw_scores(mf, tm, score_ref=c["score_ref"], band_size=c["band_size"], score_mode=dc.score_mode)
    print('  dtw_scores(all) == batch_detect scores:', np.array_equal(sc_all, scores))
    for wi, _, _ in diffs[:3]:
        sub = mf[:1, wi:wi + L + 2]  # 3 windows of one stream -> the single-stream kernels
        s3, _, a3 = ctx.dtw_scores(sub, tm, score_ref=c["score_ref"], band_size=c["band_size"], score_mode=dc.score_mode)
        print('  window', wi, 'offline', scores[0, wi], 'small-call', s3[0, 0])
