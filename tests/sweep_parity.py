"""Randomised parity sweep (test infrastructure): random wakeword references, detector configs and streams through the
device path (rp_batch_detect_fmt, rp_stream_batch_process) against the oracle's chunked detector.

Every case draws K, T, ragged template lengths, band size, score mode, thresholds, min_scores, eager, score_ref, an
optional averaged template, the VAD mode, the sample format, S streams of noise with utterances planted in them, and a
ragged tail.  The bar is the one of tests/test_gpu_parity.py: detections exact in (chunk, counter), scores within 1e-5
relative.  A case whose decision hangs on a score closer than 1e-5 to a threshold is counted as a tie, not a failure.

    python tests/sweep_parity.py --cases 300 --seed 1        # on the GPU box
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if HERE not in sys.path:
    sys.path.insert(0, HERE)

MODES = ["average", "max", "median", "p25", "p50", "p75", "p80", "p90", "p95"]


def _utterance(rng, n):
    """A few drifting partials under a random envelope + a little noise: MFCC trajectories that DTW can lock onto."""
    t = np.arange(n, dtype=np.float64) / 16000.0
    x = np.zeros(n)
    for _ in range(int(rng.integers(2, 6))):
        f0, f1 = rng.uniform(120, 3500, 2)
        ph = 2 * np.pi * np.cumsum(np.linspace(f0, f1, n)) / 16000.0
        env = np.interp(t, np.linspace(0, t[-1], 8), rng.uniform(0.0, 1.0, 8))
        x += rng.uniform(0.05, 0.3) * env * np.sin(ph + rng.uniform(0, 6.28))
    x += rng.standard_normal(n) * 0.003
    return x.astype(np.float32)


def make_case(rng, extreme=False, mfma=False, ragged=False):
    from oracle import rp_oracle as orc
    # (mfcc size 1 is left to the MFCC sweep: with one coefficient every cosine is +-1, window scores repeat exactly and
    # which of two equal-scoring windows a detection reports -- its avg_score -- hangs on the last bit)
    K = int(rng.choice([5, 5, 5, 16, 3, 8, 12]))  # (2 coefficients: an avg_score 1.5e-5 off on 7-bit audio; the extreme sweeps keep K = 2)
    T = int(rng.integers(1, 9))
    if rng.random() < 0.3:
        lens = np.full(T, int(rng.integers(20, 100)))
    else:
        lens = rng.integers(20, 110, size=T)
    if extreme:  # template sets at the edges too: 1-3 frame templates, 20 templates of one length, 40 coefficients
        K = int(rng.choice([5, 16, 40, 2]))
        T = int(rng.choice([1, 2, 9, 20]))
        lens = np.full(T, int(rng.integers(20, 60))) if rng.random() < 0.4 else rng.choice([1, 2, 3, 5, 8, 30, 64, 65, 128], size=T)
    if mfma:  # the shapes dtw_mfma_kernel takes: mfcc_size 5, chunks of 3..8 same-length templates (plus, sometimes, a few others)
        K = int(rng.choice([5, 5, 5, 16, 13]))  # 13 / 16: dtw_mfma_wide_kernel, when every length occurs at least three times
        T = int(rng.integers(3, 17))
        lens = np.full(T, int(rng.integers(12, 110)))
        if rng.random() < (0.4 if K == 5 else 0.15):
            lens[: int(rng.integers(1, 4))] = rng.integers(12, 110)
        elif K != 5 and T >= 6 and rng.random() < 0.4:
            lens[: T // 2] = int(rng.integers(12, 110))  # two lengths, three or more templates each
        if K == 5 and rng.random() < 0.12:  # round 5: four or five chunks of one length -- dtw_mfma_group_kernel (run_sweep forces it below its size rule)
            T = int(rng.integers(32, 45))
            lens = np.full(T, int(rng.integers(12, 108)))
            if rng.random() < 0.4:
                lens[: int(rng.integers(1, 4))] = rng.integers(12, 110)
    if ragged:  # the shapes dtw_ragged_kernel takes (opt-in, RP_ARITH_FAST_SPLIT + ragged_matrix): mfcc_size 5, templates whose lengths all differ (sometimes one pair)
        K = 5
        T = int(rng.integers(1, 10))
        lens = rng.choice(np.arange(16, 111), size=T, replace=False)
        if T >= 3 and rng.random() < 0.2:
            lens[1] = lens[0]
    utts = [_utterance(rng, 480 * ((int(L) + 3 + 2) // 3)) for L in lens]
    templates = [orc.normalize(orc.mfcc_stream(u, K))[:int(L)] for u, L in zip(utts, lens)]
    avg = None
    if rng.random() < 0.5 and not ragged:
        avg = templates[int(rng.integers(T))][:int(rng.integers(min(10, int(lens.max())), int(lens.max()) + 1))]
        if rng.random() < 0.3:  # an averaged template longer than every sample template (window shorter than it)
            avg = orc.normalize(orc.mfcc_stream(_utterance(rng, 480 * 50), K))[:int(lens.max()) + int(rng.integers(1, 20))]
    cfg = dict(threshold=float(rng.uniform(0.3, 0.58)), avg_threshold=float(rng.choice([0.0, 0.0, rng.uniform(0.1, 0.45)])),
               min_scores=int(rng.integers(1, 7)), eager=bool(rng.random() < 0.3), score_ref=float(rng.uniform(0.15, 0.3)),
               band_size=int(rng.integers(1, 9)), score_mode=str(rng.choice(MODES)),
               vad_mode=[None, None, None, "easy", "medium", "hard"][int(rng.integers(6))])
    if mfma:
        # round 4: score_ref down to the matrix-core kernels' floor (0.05; the relative error of a score grows like 1 / score_ref)
        cfg.update(band_size=int(rng.choice([3, 4, 5, 5, 5])) if K == 5 else 5, score_ref=float(rng.uniform(0.05, 0.3)))
        if rng.random() < 0.5:  # detect-only calls in ScoreMode::Max abandon hopeless DTWs: half of the cases take that path too
            cfg.update(score_mode="max")
    if ragged:  # every window against every template (no gate), the kernel's bands and its score_ref floor; half of the cases detect-only + abandon
        cfg.update(band_size=int(rng.choice([3, 4, 5, 5, 5])), score_ref=float(rng.uniform(0.1, 0.3)), avg_threshold=0.0)
        if rng.random() < 0.5:
            cfg.update(score_mode="max")
    if extreme:  # see make_api_case
        cfg.update(band_size=int(rng.choice([0, 1, 2, 15, 40, 120])), min_scores=int(rng.choice([0, 1, 2, 50])),
                   threshold=float(rng.choice([0.0, 1e-6, 0.3, 0.99, 1.5, -0.5])),
                   avg_threshold=float(rng.choice([0.0, 0.0, -1.0, 0.3, 2.0])),
                   score_ref=float(rng.choice([0.01, 0.05, 0.22, 5.0])))  # far smaller: (cost - ref) / ref turns 1e-8 of cost into percents
    S = int(rng.integers(1, 5))
    n_chunks = int(rng.integers(70 if ragged else 45, 150))   # (ragged: at least 64 windows per stream)
    N = 480 * n_chunks + int(rng.choice([0, 0, rng.integers(1, 480)]))
    pcm = (rng.standard_normal((S, N)) * rng.uniform(0.0005, 0.02)).astype(np.float32)
    for s in range(S):
        for _ in range(int(rng.integers(0, 5))):
            u = utts[int(rng.integers(T))]
            u = u * np.float32(rng.uniform(0.7, 1.2)) + (rng.standard_normal(len(u)) * rng.uniform(0.0, 0.002)).astype(np.float32)
            at = int(rng.integers(0, max(1, N - len(u))))
            seg = pcm[s, at:at + len(u)]
            seg += u[:len(seg)]
    if rng.random() < 0.15:
        pcm[0, : N // 2] = 0.0  # digital silence: constant features, zero vectors in the cosine
    fmt = str(rng.choice(["f32", "f32", "i16", "i16", "i8", "i32"]))
    if fmt != "f32":
        info = np.iinfo({"i16": np.int16, "i8": np.int8, "i32": np.int32}[fmt])
        pcm = np.clip(np.round(pcm.astype(np.float64) * (8.0 if fmt == "i8" else 1.0) * info.max), info.min, info.max).astype(info.dtype)
    return dict(K=K, templates=templates, avg=avg, cfg=cfg, pcm=pcm, chunks_per_call=int(rng.integers(1, 6)))


def oracle_detections(case):
    from oracle import rp_oracle as orc
    c = case["cfg"]
    out = []
    ww = {"name": "w", "samples_features": {"t%d" % i: t for i, t in enumerate(case["templates"])},
          "avg_features": case["avg"], "threshold": None, "avg_threshold": None, "rms_level": 0.0}
    for s in range(case["pcm"].shape[0]):
        d = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"],
                         score_ref=c["score_ref"], band_size=c["band_size"], score_mode=c["score_mode"], vad_mode=c["vad_mode"])
        d.add_ref(ww)
        x = case["pcm"][s]
        got = []
        for i in range(0, len(x) - 479, 480):
            r = d.process_i16(x[i:i + 480]) if x.dtype == np.int16 else d.process_f32(_to_f32(x[i:i + 480]))
            if r is not None:
                got.append((i // 480, int(r["counter"]), float(r["score"]), float(r["avg_score"])))
        out.append(got)
    return out


def device_detections(ra, ctx, case):
    c = case["cfg"]
    dc = ra.DetectorConfig()
    dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
    dc.score_ref, dc.band_size = c["score_ref"], c["band_size"]
    dc.score_mode = {m: getattr(ra.ScoreMode, m.capitalize()) for m in MODES}[c["score_mode"]]
    dc.vad_mode = {None: None, "easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[c["vad_mode"]]
    tm = ra.Templates(ctx, case["templates"], avg=case["avg"])
    pcm = case["pcm"]
    det, n_det, scores, agg = ctx.batch_detect(pcm, tm, dc, max_det=kMaxDet, want_scores=True)
    # the same call without the per-window arrays: with an averaged template and avg_threshold != 0 this takes the gate-skip
    # path where a register kernel exists (windows below avg_threshold are not compared with the sample templates) -- the
    # detections must be the same bits
    det2, n_det2 = ctx.batch_detect(pcm, tm, dc, max_det=kMaxDet)
    assert np.array_equal(n_det, n_det2) and det.tobytes() == det2.tobytes(), "gate-skip path differs from the full path"
    # n_det counts every detection; the first max_det are stored
    offline = [[(int(det[s][j]["frame"]) // 3 + 1, int(det[s][j]["counter"]), float(det[s][j]["score"]), float(det[s][j]["avg_score"]))
                for j in range(min(int(n_det[s]), kMaxDet))] + [None] * max(0, int(n_det[s]) - kMaxDet) for s in range(pcm.shape[0])]
    # the same streams a few chunks per call
    sb = ra.StreamBatch(ctx, tm, dc, pcm.shape[0], max_chunks_per_call=case["chunks_per_call"])
    live = [[] for _ in range(pcm.shape[0])]
    step = 480 * case["chunks_per_call"]
    n = (pcm.shape[1] // 480) * 480
    for i in range(0, n, step):
        d, nd = sb.process(np.ascontiguousarray(pcm[:, i:min(i + step, n)]), max_det=16)  # at most one detection per frame
        for s in range(pcm.shape[0]):
            live[s] += [(int(d[s][j]["frame"]) // 3 + 1, int(d[s][j]["counter"]), float(d[s][j]["score"]), float(d[s][j]["avg_score"]))
                        for j in range(nd[s])]
    return offline, live, agg


kMaxDet = 32


def _to_f32(v):
    """`v as f32 / T::MAX as f32` (src/audio/audio_types.rs:98-137)"""
    scale = {np.dtype(np.int8): 127.0, np.dtype(np.int16): 32767.0, np.dtype(np.int32): 2147483648.0}.get(v.dtype)
    return v if scale is None else v.astype(np.float32) / np.float32(scale)


def _same(a, b, rtol):
    if len(a) != len(b):
        return False
    for x, y in zip(a[:kMaxDet], b[:kMaxDet]):
        if x[0] != y[0] or x[1] != y[1]:
            return False
        for u, v in ((x[2], y[2]), (x[3], y[3])):
            if abs(u - v) > rtol * max(abs(v), 1e-30):
                return False
    return True


def run_sweep(ra, ctx, n_cases, seed, verbose=False, extreme=False, mfma=False, ragged=False, arith=None):
    """-> (cases, detections compared, ties skipped); raises AssertionError with the case number on a mismatch.
    arith: rp_ctx_set_arithmetic for the device calls ("f32_matrix" = the library default, "strict_f32", "fast_split"); None = the context's.
    ragged: the opt-in matrix-core kernel for templates of unequal length scores the offline calls (fast_split + ragged_matrix; the family asserts
    that it ran); live-stream batches keep the register kernels, so live vs offline is counters exact + scores within 4e-6 there."""
    if ragged:
        arith = "fast_split"
    old_arith = ctx.get_arithmetic()
    if arith is not None:
        ctx.set_arithmetic(arith, ragged)
    total = ties = 0
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 55, ci] if ragged else [seed, 77, ci] if mfma else [seed, ci])
        case = make_case(rng, extreme=extreme, mfma=mfma, ragged=ragged)
        ref = oracle_detections(case)
        if ragged:
            ctx.dtw_kernels()
        if mfma:
            os.environ["RP_DTW_GROUP"] = "2"   # (tuning knob) references with four chunks of one length: the group form whatever the launch size (same bits: live == offline stays exact; fast_split only)
        try:
            offline, live, agg = device_detections(ra, ctx, case)
        except BaseException:
            ctx.set_arithmetic(*old_arith)
            raise
        finally:
            if mfma:
                del os.environ["RP_DTW_GROUP"]
        if ragged:
            assert "dtw_ragged_kernel" in ctx.dtw_kernels(), "ragged sweep seed %d case %d: the kernel did not run (lens %r cfg %r)" % (
                seed, ci, [len(t) for t in case["templates"]], case["cfg"])
        # extreme parameters: thresholds <= 0 let scores of 1e-4 .. 1e-20 through, whose relative error is the whole cost error
        # divided by score_ref -- two f32 evaluations of the same DTW already differ by that (score_ref 0.05, 2-coefficient frames,
        # generic f32 kernel: 1.1e-5; score_ref 0.01: 3e-4).  The decisions (chunk, counter) stay exact; the scores are compared at
        # 1e-5 from score_ref 0.2 up (round 4; it was 1e-3 for the whole family), at 1e-5 x 0.22 / score_ref down to 0.05, at 1e-3 below
        # Degenerate shapes of that family (2-coefficient frames, templates of 1..4 frames) stay at 1e-3: a window of three nearly equal
        # frames minus its own mean is rounding noise, and the 1e-5 the two MFCC implementations may differ by turns its direction
        # (seed 2023 case 138: K 2, one 3-frame template, 1.2e-4 on one score at the default score_ref).
        sr = case["cfg"]["score_ref"]
        degenerate = case["K"] <= 2 or min(len(t) for t in case["templates"]) < 5
        tol = 1e-5 if not extreme else (1e-3 if (degenerate or sr < 0.05) else 1e-5 * max(1.0, 0.22 / sr))
        live_tol = 4e-6 if ragged else 0.0
        ok = all(_same(o, r, tol) for o, r in zip(offline, ref)) and all(_same(l, o, live_tol) for l, o in zip(live, offline))
        if not ok:
            thr = case["cfg"]["threshold"]
            near = agg.size and np.min(np.abs(agg - np.float32(thr))) < 1e-5 * thr
            if near and (ragged or all(_same(l, o, 0.0) for l, o in zip(live, offline))):
                ties += 1
                continue
            ctx.set_arithmetic(*old_arith)
            raise AssertionError("sweep seed %d case %d: cfg %r K %d lens %r\noracle  %r\noffline %r\nlive    %r" % (
                seed, ci, case["cfg"], case["K"], [len(t) for t in case["templates"]], ref, offline, live))
        total += sum(len(r) for r in ref)
        if verbose and ci % 20 == 0:
            print("case %d ok, %d detections so far" % (ci, total), flush=True)
    ctx.set_arithmetic(*old_arith)
    return n_cases, total, ties


# ----------------------------------------------------------------------------------------------------------------
# Live-stream batches with `Rustpotter::reset` of single streams (rp_stream_batch_reset) at random call boundaries, against
# the oracle's detector reset at the same chunks.
def run_live_reset_sweep(ra, ctx, n_cases, seed, verbose=False):
    from oracle import rp_oracle as orc
    total = 0
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 22, ci])
        case = make_case(rng)
        c, pcm = case["cfg"], case["pcm"]
        S, cpc = pcm.shape[0], case["chunks_per_call"]
        n = (pcm.shape[1] // 480) * 480
        calls = list(range(0, n, 480 * cpc))
        resets = {}  # call index -> streams reset in front of it (-1: all)
        for _ in range(int(rng.integers(1, 4))):
            resets.setdefault(int(rng.integers(len(calls))), []).append(int(rng.integers(-1, S)))
        dc = ra.DetectorConfig()
        dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
        dc.score_ref, dc.band_size = c["score_ref"], c["band_size"]
        dc.score_mode = getattr(ra.ScoreMode, c["score_mode"].capitalize())
        dc.vad_mode = {None: None, "easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[c["vad_mode"]]
        tm = ra.Templates(ctx, case["templates"], avg=case["avg"])
        sb = ra.StreamBatch(ctx, tm, dc, S, max_chunks_per_call=cpc)
        ww = {"name": "w", "samples_features": {"t%d" % i: t for i, t in enumerate(case["templates"])},
              "avg_features": case["avg"], "threshold": None, "avg_threshold": None, "rms_level": 0.0}
        dets = []
        for s in range(S):
            d = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"],
                             score_ref=c["score_ref"], band_size=c["band_size"], score_mode=c["score_mode"], vad_mode=c["vad_mode"])
            d.add_ref(ww)
            dets.append(d)
        ref = [[] for _ in range(S)]
        live = [[] for _ in range(S)]
        for k, i in enumerate(calls):
            for who in resets.get(k, []):
                sb.reset(who)
                for s in (range(S) if who < 0 else [who]):
                    dets[s].reset()
            piece = np.ascontiguousarray(pcm[:, i:min(i + 480 * cpc, n)])
            dd, nd = sb.process(piece, max_det=16)
            for s in range(S):
                live[s] += [(int(dd[s][j]["frame"]) // 3 + 1, int(dd[s][j]["counter"]), float(dd[s][j]["score"]), float(dd[s][j]["avg_score"]))
                            for j in range(nd[s])]
                for q in range(piece.shape[1] // 480):
                    x = piece[s, 480 * q:480 * (q + 1)]
                    r = dets[s].process_i16(x) if x.dtype == np.int16 else dets[s].process_f32(_to_f32(x))
                    if r is not None:
                        ref[s].append((i // 480 + q, int(r["counter"]), float(r["score"]), float(r["avg_score"])))
        ok = all(_same(l, r, 1e-5) for l, r in zip(live, ref))
        assert ok, "live reset sweep seed %d case %d: cfg %r resets %r cpc %d\noracle %r\nlive   %r" % (seed, ci, c, resets, cpc, ref, live)
        total += sum(len(r) for r in ref)
        if verbose and ci % 20 == 0:
            print("live-reset case %d ok, %d detections so far" % (ci, total), flush=True)
    return n_cases, total


# ----------------------------------------------------------------------------------------------------------------
# Live-stream batches behind the resampler (rp_stream_batch_set_input: 8-48 kHz, 1-2 interleaved channels, i16 / f32)
# against the oracle's resampler + detector, stream by stream.
def run_live_rate_sweep(ra, ctx, n_cases, seed, verbose=False):
    """-> (cases, detections compared with the oracle, cases that differ from the oracle by a near-tie decision)"""
    from oracle import rp_oracle as orc
    total, ties = 0, []
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 23, ci])
        case = make_case(rng)
        c = case["cfg"]
        rate = int(rng.choice([48000, 48000, 8000, 32000, 44100, 24000, 96000, 22050, 11025]))  # the last two: 40 ms frames
        ch = int(rng.choice([1, 2]))
        cpc = int(rng.integers(1, 4))
        fi, fo = ra.resampler_frame_lengths(rate)
        base = _to_f32(case["pcm"])
        # no digital silence here: behind a resampler silence turns into 1e-9 ringing whose log-mel features are chaotic in
        # any implementation (the reason the reference's 48 kHz model goldens cannot be pinned either)
        quiet = np.abs(base) < 1e-7
        base = np.where(quiet, (rng.standard_normal(base.shape) * 1e-3).astype(np.float32), base)
        S = base.shape[0]
        n16 = (base.shape[1] // 480) * 480
        # any signal at the input rate will do: stretch the 16 kHz case
        m = int(n16 * rate / 16000) // fi * fi
        x = np.stack([np.interp(np.arange(m) * (16000.0 / rate), np.arange(n16), base[s, :n16]) for s in range(S)]).astype(np.float32)
        if rng.random() < 0.5:
            raw = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
            dec = raw.astype(np.float32) / np.float32(32767.0)
        else:
            raw, dec = x, x
        inter = raw if ch == 1 else np.stack([raw, np.roll(raw, 5, axis=1)], axis=2).reshape(S, m * 2)
        dc = ra.DetectorConfig()
        dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
        dc.score_ref, dc.band_size = c["score_ref"], c["band_size"]
        dc.score_mode = getattr(ra.ScoreMode, c["score_mode"].capitalize())
        dc.vad_mode = {None: None, "easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[c["vad_mode"]]
        tm = ra.Templates(ctx, case["templates"], avg=case["avg"])
        sb = ra.StreamBatch(ctx, tm, dc, S, max_chunks_per_call=cpc, sample_rate=rate, channels=ch)
        assert sb.samples_per_chunk == fi * ch
        ww = {"name": "w", "samples_features": {"t%d" % i: t for i, t in enumerate(case["templates"])},
              "avg_features": case["avg"], "threshold": None, "avg_threshold": None, "rms_level": 0.0}
        ref, live = [], [[] for _ in range(S)]
        for s in range(S):
            d = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"],
                             score_ref=c["score_ref"], band_size=c["band_size"], score_mode=c["score_mode"], vad_mode=c["vad_mode"])
            d.add_ref(ww)
            rs = orc.Resampler(rate)
            got = []
            for k in range(m // fi):
                r = d.process_resampled(rs, dec[s, k * fi:(k + 1) * fi])
                if r is not None:
                    got.append((k, int(r["counter"]), float(r["score"]), float(r["avg_score"])))
            ref.append(got)
        # frames per input frame: fo / 160 (3 for 480-sample output frames, 4 for 640)
        fpf = fo // 160
        for i in range(0, m // fi, cpc):
            nf = min(cpc, m // fi - i)
            piece = np.ascontiguousarray(inter[:, i * fi * ch:(i + nf) * fi * ch])
            dd, nd = sb.process(piece, max_det=16)
            for s in range(S):
                live[s] += [(int(dd[s][j]["frame"]), int(dd[s][j]["counter"]), float(dd[s][j]["score"]), float(dd[s][j]["avg_score"]))
                            for j in range(nd[s])]
        # the oracle reports the input frame in which a detection was returned; the batch reports the 10 ms frame
        live = [[((f + 3) // fpf, cnt, sc, av) for f, cnt, sc, av in l] for l in live]
        # (1) the per-call resampler state: the live batch must equal, bit for bit, resampling the whole recording on
        # the device and running the offline batch on it
        y = ctx.resample(np.ascontiguousarray(inter), rate, channels=ch)
        det, n_det = ctx.batch_detect(y, tm, dc, max_det=kMaxDet)
        off = [[((int(det[s][j]["frame"]) + 3) // fpf, int(det[s][j]["counter"]), float(det[s][j]["score"]), float(det[s][j]["avg_score"]))
                for j in range(min(int(n_det[s]), kMaxDet))] + [None] * max(0, int(n_det[s]) - kMaxDet) for s in range(S)]
        # (40 ms input frames: a detection drops the rest of ITS input frame and the offline pass over 16 kHz audio cuts
        # frames of 30 ms, so the two legitimately part ways after the first detection of a stream)
        assert fpf != 3 or all(_same(l, o, 0.0) for l, o in zip(live, off)), "live rate sweep seed %d case %d: rate %d ch %d cpc %d %s: live != offline\n%r\n%r" % (
            seed, ci, rate, ch, cpc, raw.dtype, live, off)
        # (2) against the oracle's resampler + detector.  Behind the resampler the audio differs by up to 4e-6 of the
        # peak, so a VAD or threshold comparison that is nearly a tie can go the other way: such cases are counted
        # (the first run of this sweep had 5 in 300, all of them on the stream that carried digital silence)
        if all(_same(l, r, 1e-4) for l, r in zip(live, ref)):
            total += sum(len(r) for r in ref)
        else:
            ties.append((ci, rate, c["vad_mode"], c["avg_threshold"], ref, live))
            if verbose:
                print("near-tie case %d: rate %d vad %r avg_threshold %.3g\n  oracle %r\n  live   %r" % (ci, rate, c["vad_mode"], c["avg_threshold"], ref, live), flush=True)
        if verbose and ci % 20 == 0:
            print("live-rate case %d ok, %d detections so far, %d near-tie cases" % (ci, total, len(ties)), flush=True)
    assert len(ties) <= max(2, n_cases // 20), "live rate sweep seed %d: %d of %d cases differ from the oracle: %r" % (seed, len(ties), n_cases, ties[:3])
    return n_cases, total, len(ties)


# ----------------------------------------------------------------------------------------------------------------
# The single-stream drop-in API (`Rustpotter`, src/detector.rs) chunk by chunk against the oracle's detector: several
# wakewords with their own thresholds, gain normaliser / band-pass, VAD, resets in mid-stream, i16 / f32 input,
# mono / stereo, 16 kHz or 48 kHz (resampler in front).
def make_api_case(rng, extreme=False, models=False):
    """extreme: detector parameters from the edges of (and outside) their sensible ranges -- band 0 / wider than the
    templates, thresholds <= 0 or > 1, min_scores 0, tiny / huge score_ref -- where the reference's arithmetic is still
    defined (all paths empty -> +inf cost -> score 0, everything above a threshold <= 0, ...)."""
    from oracle import rp_oracle as orc
    K = int(rng.choice([5, 5, 16, 8]))
    wakewords, utts = [], []
    for wi in range(int(rng.integers(1, 4))):
        T = int(rng.integers(1, 6))
        lens = rng.integers(20, 100, size=T)
        us = [_utterance(rng, 480 * ((int(L) + 5) // 3)) for L in lens]
        utts += us
        feats = {"s%d.wav" % i: orc.normalize(orc.mfcc_stream(u, K))[:int(L)] for i, (u, L) in enumerate(zip(us, lens))}
        avg = None
        if rng.random() < 0.5:
            avg = feats["s%d.wav" % int(rng.integers(T))][:int(rng.integers(10, int(lens.max()) + 1))]
        wakewords.append({"name": "ww%d" % wi, "samples_features": feats, "avg_features": avg,
                          "threshold": None if rng.random() < 0.6 else float(rng.uniform(0.3, 0.55)),
                          "avg_threshold": None if rng.random() < 0.6 else float(rng.choice([0.0, rng.uniform(0.1, 0.4)])),
                          "rms_level": float(rng.uniform(0.01, 0.2))})
    cfg = dict(threshold=float(rng.uniform(0.25, 0.52)), avg_threshold=float(rng.choice([0.0, 0.0, rng.uniform(0.1, 0.45)])),
               min_scores=int(rng.integers(1, 7)), eager=bool(rng.random() < 0.3), score_ref=float(rng.uniform(0.15, 0.3)),
               band_size=int(rng.integers(1, 9)), score_mode=str(rng.choice(MODES)),
               vad_mode=[None, None, None, "easy", "medium", "hard"][int(rng.integers(6))],
               gain_normalizer=bool(rng.random() < 0.35), gain_ref=None if rng.random() < 0.5 else float(rng.uniform(0.01, 0.1)),
               min_gain=float(rng.uniform(0.1, 0.5)), max_gain=float(rng.uniform(1.0, 3.0)),
               band_pass=bool(rng.random() < 0.3), low_cutoff=float(rng.uniform(60, 200)), high_cutoff=float(rng.uniform(300, 3000)))
    if extreme:
        cfg.update(band_size=int(rng.choice([0, 1, 2, 15, 40, 120])), min_scores=int(rng.choice([0, 1, 2, 50])),
                   threshold=float(rng.choice([0.0, 1e-6, 0.3, 0.99, 1.5, -0.5])),
                   avg_threshold=float(rng.choice([0.0, 0.0, -1.0, 0.3, 2.0])),
                   score_ref=float(rng.choice([0.01, 0.05, 0.22, 5.0])))  # far smaller: (cost - ref) / ref turns 1e-8 of cost into percents
    n_chunks = int(rng.integers(50, 160))
    x = (rng.standard_normal(480 * n_chunks) * rng.uniform(0.0005, 0.02)).astype(np.float32)
    for _ in range(int(rng.integers(2, 7))):
        u = utts[int(rng.integers(len(utts)))] * np.float32(rng.uniform(0.7, 1.2))
        at = int(rng.integers(0, max(1, len(x) - len(u))))
        x[at:at + len(u)] += u[:len(x) - at]
    rate = int(rng.choice([48000, 48000, 22050, 11025, 8000, 44100, 32000])) if rng.random() < 0.3 else 16000
    if rate != 16000:  # any signal at that rate will do: the encoder in front of the detector is what is compared
        m = int(len(x) * rate / 16000)
        x = np.interp(np.arange(m) * (16000.0 / rate), np.arange(len(x)), x).astype(np.float32)
        x = np.where(np.abs(x) < 1e-7, np.float32(1e-3) * rng.standard_normal(m).astype(np.float32), x)  # no digital silence behind a resampler
    fmt = str(rng.choice(["f32", "f32", "i16", "i16", "i8", "i32"]))  # the reference's four `Sample` types
    if fmt != "f32":
        info = np.iinfo({"i16": np.int16, "i8": np.int8, "i32": np.int32}[fmt])
        x = np.clip(np.round(x.astype(np.float64) * (8.0 if fmt == "i8" else 1.0) * info.max), info.min, info.max).astype(info.dtype)  # (i8: louder, 7 bits)
    resets = sorted(set(int(v) for v in rng.integers(0, n_chunks, size=int(rng.integers(0, 3)))))
    # update_detector_config / update_filters_config in mid-stream (src/detector.rs:255-289)
    updates = {}
    for _ in range(int(rng.integers(0, 3))):
        at = int(rng.integers(1, n_chunks))
        if rng.random() < 0.5:
            updates[at] = ("detector", dict(threshold=float(rng.uniform(0.25, 0.52)), avg_threshold=float(rng.choice([0.0, rng.uniform(0.1, 0.45)])),
                                            min_scores=int(rng.integers(1, 7)), eager=bool(rng.random() < 0.3), score_ref=float(rng.uniform(0.15, 0.3)),
                                            band_size=int(rng.integers(1, 9)), score_mode=str(rng.choice(MODES)),
                                            vad_mode=[None, None, "easy", "medium", "hard"][int(rng.integers(5))]))
        else:
            updates[at] = ("filters", dict(gain_normalizer=bool(rng.random() < 0.5), gain_ref=None if rng.random() < 0.5 else float(rng.uniform(0.01, 0.1)),
                                           min_gain=float(rng.uniform(0.1, 0.5)), max_gain=float(rng.uniform(1.0, 3.0)),
                                           band_pass=bool(rng.random() < 0.5), low_cutoff=float(rng.uniform(60, 200)),
                                           high_cutoff=float(rng.uniform(300, 3000))))
    # a wakeword model next to the references now and then (run_wakeword_detectors picks the best score across both kinds)
    if models and rng.random() < 0.25:
        fr = int(rng.integers(30, 110))
        mt = int(rng.integers(4))
        hidden = {0: [fr // 15], 1: [fr // 6, (fr // 6) // 2], 2: [fr // 3, fr // 6], 3: [(fr // 3) * 2, fr // 6]}[mt]
        labels = ["none", "gamma", "delta"][:int(rng.integers(2, 4))]
        dims = [fr * K] + hidden + [len(labels)]
        weights = {}
        for i in range(len(dims) - 1):
            weights["ln%d.weight" % (i + 1)] = (rng.standard_normal((dims[i + 1], dims[i])) * (1.5 / np.sqrt(dims[i]))).astype(np.float32)
            weights["ln%d.bias" % (i + 1)] = (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)
        wakewords.append({"kind": "model", "name": "model", "labels": labels, "train_size": fr, "mfcc_size": K,
                          "m_type": ["Tiny", "Small", "Medium", "Large"][mt], "weights": weights, "rms_level": float(rng.uniform(0.01, 0.2))})
    # remove_wakeword / add_wakeword in mid-stream (src/detector.rs:144-202, on_wakeword_change :328-346): the window is
    # NOT reset, so it can be longer than the largest remaining wakeword needs, or has to grow for a longer new one
    n_initial = len(wakewords)
    ww_events = {}
    if rng.random() < 0.4:
        for wi in range(n_initial, n_initial + int(rng.integers(0, 3))):  # extra wakewords that join later
            T = int(rng.integers(1, 4))
            lens = rng.integers(20, 120, size=T)
            us = [_utterance(rng, 480 * ((int(L) + 5) // 3)) for L in lens]
            feats = {"s%d.wav" % i: orc.normalize(orc.mfcc_stream(u, K))[:int(L)] for i, (u, L) in enumerate(zip(us, lens))}
            wakewords.append({"name": "ww%d" % wi, "samples_features": feats, "avg_features": None, "threshold": None,
                              "avg_threshold": None, "rms_level": float(rng.uniform(0.01, 0.2))})
        for _ in range(int(rng.integers(1, 4))):
            ww_events[int(rng.integers(1, n_chunks))] = (str(rng.choice(["remove", "add"])), int(rng.integers(len(wakewords))))
    return dict(K=K, wakewords=wakewords, n_initial=n_initial, ww_events=ww_events, cfg=cfg, x=x, rate=rate,
                channels=int(rng.choice([1, 1, 2])), resets=resets, updates=updates)


def run_api_sweep(ra, n_cases, seed, verbose=False, extreme=False):
    from oracle import rp_oracle as orc
    import rpw_py
    total = 0
    for ci in range(n_cases):
        case = make_api_case(np.random.default_rng([seed, 77, ci]), extreme=extreme, models=True)
        c = case["cfg"]
        d = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"],
                         score_ref=c["score_ref"], band_size=c["band_size"], score_mode=c["score_mode"], vad_mode=c["vad_mode"],
                         gain_normalizer=c["gain_normalizer"], gain_ref=c["gain_ref"], min_gain=c["min_gain"], max_gain=c["max_gain"],
                         band_pass=c["band_pass"], low_cutoff=c["low_cutoff"], high_cutoff=c["high_cutoff"])
        rc = ra.RustpotterConfig.default()
        x = case["x"]
        rc.fmt.sample_rate, rc.fmt.channels = case["rate"], case["channels"]
        rc.fmt.sample_format = {np.dtype(np.int8): ra.SampleFormat.I8, np.dtype(np.int16): ra.SampleFormat.I16,
                                np.dtype(np.int32): ra.SampleFormat.I32, np.dtype(np.float32): ra.SampleFormat.F32}[x.dtype]
        # `v as f32 / T::MAX as f32` (src/audio/audio_types.rs:98-137) for the oracle's f32 entry points
        to_f32 = {np.dtype(np.int8): lambda v: v.astype(np.float32) / np.float32(127.0), np.dtype(np.int16): lambda v: v.astype(np.float32) / np.float32(32767.0),
                  np.dtype(np.int32): lambda v: v.astype(np.float32) / np.float32(2147483648.0), np.dtype(np.float32): lambda v: v}[x.dtype]
        # a third of the cases hand the chunks over as bytes (process_bytes, src/detector.rs:234-243) in either byte order
        byte_order = [None, None, "<", ">"][int(np.random.default_rng([seed, 78, ci]).integers(4))]
        if byte_order:
            rc.fmt.endianness = ra.Endianness.Little if byte_order == "<" else ra.Endianness.Big
        dc = rc.detector
        dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
        dc.score_ref, dc.band_size = c["score_ref"], c["band_size"]
        dc.score_mode = getattr(ra.ScoreMode, c["score_mode"].capitalize())
        dc.vad_mode = {None: None, "easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[c["vad_mode"]]
        g, b = rc.filters.gain_normalizer, rc.filters.band_pass
        g.enabled, g.gain_ref, g.min_gain, g.max_gain = c["gain_normalizer"], c["gain_ref"], c["min_gain"], c["max_gain"]
        b.enabled, b.low_cutoff, b.high_cutoff = c["band_pass"], c["low_cutoff"], c["high_cutoff"]
        rp = ra.Rustpotter.new(rc)
        active = []  # names in insertion order = the oracle's indices

        def add(w):
            if w.get("kind") == "model":
                d.add_model(w)
                rp.add_wakeword_from_buffer(w["name"], rpw_py.dump_rpw_model(w["labels"], w["train_size"], w["mfcc_size"], w["m_type"],
                                                                              w["weights"], w["rms_level"]))
            else:
                d.add_ref(w)
                rp.add_wakeword_from_buffer(w["name"], rpw_py.dump_rpw_ref(w["name"], w["samples_features"], w["avg_features"],
                                                                            w["threshold"], w["avg_threshold"], w["rms_level"]))
            active.append(w["name"])

        for w in case["wakewords"][:case["n_initial"]]:
            add(w)
        rs = orc.Resampler(case["rate"]) if case["rate"] != 16000 else None
        spf = rp.get_samples_per_frame()
        per = spf // case["channels"]
        assert per == (rs.in_len if rs else 480)
        where = "api sweep seed %d case %d (%r, K %d, rate %d, ch %d, %s)" % (seed, ci, c, case["K"], case["rate"], case["channels"], x.dtype)
        for k in range(len(x) // per):
            if k in case["resets"]:
                d.reset()
                rp.reset()
            if k in case["ww_events"]:
                what, wi = case["ww_events"][k]
                w = case["wakewords"][wi]
                if what == "remove" and w["name"] in active:
                    assert rp.remove_wakeword(w["name"]) and d.remove(active.index(w["name"]))
                    active.remove(w["name"])
                elif what == "add" and w["name"] not in active:
                    add(w)
            if k in case["updates"]:
                kind, u = case["updates"][k]
                if kind == "detector":
                    d.update_detector_config(u["avg_threshold"], u["threshold"], u["min_scores"], u["eager"], u["score_ref"], u["band_size"],
                                             u["score_mode"], u["vad_mode"])
                    dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = u["avg_threshold"], u["threshold"], u["min_scores"], u["eager"]
                    dc.score_ref, dc.band_size = u["score_ref"], u["band_size"]
                    dc.score_mode = getattr(ra.ScoreMode, u["score_mode"].capitalize())
                    dc.vad_mode = {None: None, "easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[u["vad_mode"]]
                    rp.update_detector_config(dc)
                else:
                    d.update_filters_config(u["gain_normalizer"], u["gain_ref"], u["min_gain"], u["max_gain"], u["band_pass"],
                                            u["low_cutoff"], u["high_cutoff"])
                    g.enabled, g.gain_ref, g.min_gain, g.max_gain = u["gain_normalizer"], u["gain_ref"], u["min_gain"], u["max_gain"]
                    b.enabled, b.low_cutoff, b.high_cutoff = u["band_pass"], u["low_cutoff"], u["high_cutoff"]
                    rp.update_filters_config(rc.filters)
            mono = x[k * per:(k + 1) * per]
            if rs:
                ref = d.process_resampled(rs, to_f32(mono))
            else:
                ref = d.process_i16(mono) if x.dtype == np.int16 else d.process_f32(to_f32(mono))
            inter = mono if case["channels"] == 1 else np.stack([mono, mono[::-1]], axis=1).reshape(-1)
            if byte_order:
                got = rp.process_bytes(np.ascontiguousarray(inter).astype(byte_order + {1: "i1", 2: "i2"}.get(x.dtype.itemsize, "i4" if x.dtype == np.int32 else "f4")).tobytes())
            else:
                got = rp.process_samples(np.ascontiguousarray(inter))
            assert (got is None) == (ref is None), "%s chunk %d: %r vs %r" % (where, k, got, ref)
            part = rp.get_partial_detection()
            assert (-1 if part is None else part.counter) == d.state()["partial_counter"], "%s chunk %d partial" % (where, k)
            if ref is None:
                continue
            total += 1
            assert got.name == ref["name"] and got.counter == ref["counter"], "%s chunk %d: %r vs %r" % (where, k, got, ref)
            for u, v in [(got.score, ref["score"]), (got.avg_score, ref["avg_score"]), (got.gain, ref["gain"])] + \
                        [(got.scores[n], ref["scores"][n]) for n in ref["scores"]]:
                # a model's `scores` are its logits: sums of large cancelling terms, compared relative to the largest of them
                is_model = any(w.get("kind") == "model" for w in case["wakewords"])
                tol = 1e-3 if extreme else 1e-4 if is_model else 1e-5
                floor = max([abs(float(t)) for t in ref["scores"].values()] + [1.0]) if is_model else 0.0
                assert abs(float(u) - float(v)) <= tol * max(abs(float(v)), floor), "%s chunk %d: %r %r vs %r" % (where, k, got, got.scores, ref)
            assert sorted(got.scores) == sorted(ref["scores"])
        if verbose and ci % 20 == 0:
            print("api case %d ok, %d detections so far" % (ci, total), flush=True)
    return n_cases, total


# ----------------------------------------------------------------------------------------------------------------
# MFCC values (rp_mfcc_batch_fmt) against the oracle over signal kinds, mfcc sizes 1..40 and levels.  Gate: SURVEY 8d's
# 1e-5 for broadband signals at levels of -40 dB and above.  Measured outside that (4 000 cases each, --report): down to
# -80 dB the worst broadband case is 1.1e-5, down to -140 dB 1.6e-5 -- the logarithms are ~-35 there, the DCT sums
# difference terms of that size and one f32 ulp of them is already 4e-6; tones reach 6.6e-5 at any level (see below).
def run_mfcc_sweep(ra, ctx, n_cases, seed, verbose=False, strict=True, min_level_exp=-2.0):
    from oracle import rp_oracle as orc
    worst = {}
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 33, ci])
        K = int(rng.choice([5, 16, 1, 2, 7, 13, 23, 40]))
        n = 480 * int(rng.integers(2, 40)) + int(rng.choice([0, rng.integers(0, 480)]))
        kind = int(rng.integers(7))
        t = np.arange(n) / 16000.0
        if kind == 0:
            x = rng.standard_normal(n)
        elif kind == 1:
            x = rng.uniform(-1, 1, n)
        elif kind == 2:  # a few steady tones
            x = sum(np.sin(2 * np.pi * rng.uniform(50, 7900) * t + rng.uniform(0, 6.28)) for _ in range(int(rng.integers(1, 4))))
        elif kind == 3:  # speech-like
            x = _utterance(rng, n).astype(np.float64)
        elif kind == 4:  # DC offset + noise
            x = rng.uniform(-1, 1) + 0.01 * rng.standard_normal(n)
        elif kind == 5:  # sparse impulses
            x = np.zeros(n)
            x[rng.integers(0, n, size=max(1, n // 300))] = rng.uniform(-1, 1, max(1, n // 300))
        else:  # band-limited noise (moving average)
            x = np.convolve(rng.standard_normal(n + 31), np.ones(32) / 32.0, "valid")
        x = (x * 10.0 ** rng.uniform(min_level_exp, 0.5)).astype(np.float32)
        if rng.random() < 0.3:
            x = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        got = ctx.mfcc(x[None, :], K)[0]
        ref = orc.mfcc_stream(x.astype(np.float32) / np.float32(32767.0) if x.dtype == np.int16 else x, K)
        assert got.shape == ref.shape
        if ref.size:
            # K <= 5: SURVEY 8d's element-wise gate; larger K: relative to the frame's largest coefficient (the DCT sums
            # reach |60| there, and the oracle is as far from an f64 evaluation as the kernel is -- see test_gpu_parity.py)
            scale = np.maximum(np.abs(ref), 1.0) if K <= 5 else np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1.0)
            err = float(np.max(np.abs(got - ref) / scale))
            worst[kind] = max(worst.get(kind, 0.0), err)
            # tones (kinds 2, 3) put the far mel filters 60-90 dB under the peak, where the rounding noise of ANY f32 FFT
            # (the kernel's 16x15 four-step, the oracle's, rustfft's) is no longer small against the local energy: the
            # logarithm turns that into absolute differences above 1e-5.  Gates (round 4, tightened to what 20 000 cases over four
            # rounds measured): steady tones 1e-4 (worst seen 6.6e-5), speech-like signals 3e-5 (worst seen 1.7e-5 in 6 000 cases); they were 2e-4
            gate = 1e-4 if kind == 2 else 3e-5 if kind == 3 else 1e-5
            assert err <= gate or not strict, "mfcc sweep seed %d case %d: kind %d K %d n %d %s level %.3g: err %.3g" % (
                seed, ci, kind, K, n, x.dtype, float(np.max(np.abs(x.astype(np.float64)))), err)
        if verbose and ci % 50 == 0:
            print("mfcc case %d ok, worst error per signal kind so far %r" % (ci, {k: float("%.3g" % v) for k, v in sorted(worst.items())}), flush=True)
    return n_cases, worst


# ----------------------------------------------------------------------------------------------------------------
# Front-end (rp_frontend_batch: decode + GainNormalizerFilter + BandPassFilter) against the oracle, bit for bit:
# stream counts around the 64-stream workgroup, row lengths with and without 4-sample alignment (tiled / per-lane
# kernel), tails shorter than a chunk, all four sample types, filter parameters across their ranges.
def run_frontend_sweep(ra, ctx, n_cases, seed, verbose=False):
    from oracle import rp_oracle as orc
    checked = 0
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 44, ci])
        S = int(rng.choice([1, 2, 63, 64, 65, 130, int(rng.integers(1, 200))]))
        N = 480 * int(rng.integers(1, 30)) + int(rng.choice([0, 0, 4 * rng.integers(0, 120), rng.integers(0, 480)]))
        level = 10.0 ** rng.uniform(-3, 0.3)
        x = rng.standard_normal((S, N)) * level
        if rng.random() < 0.3:
            x[int(rng.integers(S)), : N // 2] = 0.0
        dt = [np.float32, np.int16, np.int8, np.int32][int(rng.integers(4))]
        if dt is np.float32:
            raw, dec = x.astype(np.float32), x.astype(np.float32)
        else:
            info = np.iinfo(dt)
            raw = np.clip(np.round(x * info.max), info.min, info.max).astype(dt)
            dec = raw.astype(np.float32) / np.float32({np.int16: 32767.0, np.int8: 127.0, np.int32: 2147483648.0}[dt])
        f = ra.FiltersConfig()
        g, b = f.gain_normalizer, f.band_pass
        g.enabled, b.enabled = bool(rng.random() < 0.7), bool(rng.random() < 0.7)
        g.gain_ref = None if rng.random() < 0.5 else float(rng.uniform(0.005, 0.3))
        g.min_gain, g.max_gain = float(rng.uniform(0.05, 1.0)), float(rng.uniform(1.0, 5.0))
        b.low_cutoff, b.high_cutoff = float(rng.uniform(20, 1000)), float(rng.uniform(1000, 7900))
        rms_ref, win = float(rng.uniform(0.005, 0.3)), int(rng.integers(1, 40))
        out, rms, gains = ctx.frontend(raw, f, rms_ref, win)
        for si in sorted(set([0, S - 1, int(rng.integers(S))])):
            ro, rr, rg = orc.frontend_stream(dec[si], gain_normalizer=g.enabled, gain_ref=g.gain_ref, min_gain=g.min_gain,
                                             max_gain=g.max_gain, rms_level_ref=rms_ref, window_size=win, band_pass=b.enabled,
                                             low_cutoff=b.low_cutoff, high_cutoff=b.high_cutoff)
            # a band-pass with far-apart cutoffs is an unstable biquad (in the reference too): both sides overflow to the same NaNs
            ok = np.array_equal(rms[si], rr) and np.array_equal(gains[si], rg) and np.array_equal(out[si], ro, equal_nan=True)
            if not ok:
                bad = np.flatnonzero(~((out[si] == ro) | (np.isnan(out[si]) & np.isnan(ro))))
                raise AssertionError("frontend sweep seed %d case %d: S %d N %d %s level %.3g stream %d, gain %r (ref %r, %g..%g, rms_ref %g, win %d) "
                                     "band-pass %r (%g..%g): rms %r gains %r out %r first diffs %r got %r want %r, gains %r vs %r" % (
                    seed, ci, S, N, np.dtype(dt).name, level, si, g.enabled, g.gain_ref, g.min_gain, g.max_gain, rms_ref, win, b.enabled,
                    b.low_cutoff, b.high_cutoff, np.array_equal(rms[si], rr), np.array_equal(gains[si], rg), np.array_equal(out[si], ro),
                    bad[:4], out[si][bad[:4]], ro[bad[:4]], gains[si][:6], rg[:6]))
            checked += 1
        if verbose and ci % 50 == 0:
            print("frontend case %d ok, %d streams compared so far" % (ci, checked), flush=True)
    return n_cases, checked


# ----------------------------------------------------------------------------------------------------------------
# Resampler (rp_resample_batch: rubato FftFixedInOut restated) against the oracle over input rates, channel counts,
# sample types, stream counts and lengths.  |d| <= 4e-6 of the peak as in tests/test_gpu_parity.py (x sqrt(fi / 1440)
# above 48 kHz).
RATES = [48000, 48000, 44100, 32000, 24000, 22050, 12000, 11025, 8000, 96000, 88200, 64000, 44000, 6000, 4000, 192000]


def run_resample_sweep(ra, ctx, n_cases, seed, verbose=False):
    from oracle import rp_oracle as orc
    worst = 0.0
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 66, ci])
        fs = int(RATES[int(rng.integers(len(RATES)))])
        fi, fo = ra.resampler_frame_lengths(fs)
        ch = int(rng.choice([1, 1, 2, 3]))
        S = int(rng.integers(1, 6))
        n = fi * int(rng.integers(1, 9)) + int(rng.choice([0, rng.integers(0, fi)]))
        t = np.arange(n)
        kind = int(rng.integers(4))
        if kind == 0:
            x = rng.uniform(-0.5, 0.5, (S, n))
        elif kind == 1:
            x = 0.3 * np.sin(2 * np.pi * rng.uniform(50, 0.45 * fs) * t / fs)[None, :] + 0.01 * rng.standard_normal((S, n))
        elif kind == 2:
            x = np.where((t // int(rng.integers(50, 900))) % 2 == 0, 0.25, -0.25)[None, :] * np.ones((S, 1))
        else:
            x = rng.standard_normal((S, n)) * 10.0 ** rng.uniform(-3, -0.5)
        dt = [np.float32, np.float32, np.int16, np.int8, np.int32][int(rng.integers(5))]
        if dt is np.float32:
            raw, dec = x.astype(np.float32), x.astype(np.float32)
        else:
            info = np.iinfo(dt)
            raw = np.clip(np.round(x * info.max), info.min, info.max).astype(dt)
            dec = raw.astype(np.float32) / np.float32({np.int16: 32767.0, np.int8: 127.0, np.int32: 2147483648.0}[dt])
        inter = raw if ch == 1 else np.stack([raw] + [np.roll(raw, k + 1, axis=1) for k in range(ch - 1)], axis=2).reshape(S, n * ch)
        got = ctx.resample(np.ascontiguousarray(inter), fs, channels=ch)
        assert got.shape == (S, (n // fi) * fo)
        for si in range(S):
            ref = orc.resample_stream(dec[si], fs)
            assert ref.shape == got[si].shape
            if ref.size:
                err = float(np.abs(got[si] - ref).max()) / max(float(np.abs(ref).max()), 1e-3)
                worst = max(worst, err)
                # above 48 kHz the matrix form sums 2 * fi > 2 880 products per output sample in f32: the rounding walk
                # grows with the square root of the depth (192 kHz: 11 520 terms, measured 4.1e-6)
                gate = 4e-6 * max(1.0, (fi / 1440.0) ** 0.5)
                assert err <= gate, "resample sweep seed %d case %d: fs %d ch %d S %d n %d %s kind %d stream %d: err %.3g" % (
                    seed, ci, fs, ch, S, n, np.dtype(dt).name, kind, si, err)
        if verbose and ci % 50 == 0:
            print("resample case %d ok, worst error so far %.3g of the peak" % (ci, worst), flush=True)
    return n_cases, worst


# ----------------------------------------------------------------------------------------------------------------
# Several wakewords in the batched detector (rp_batch_detect_multi) against the oracle's detector holding the same
# wakewords (run_wakeword_detectors, src/detector.rs:433-447), per-wakeword threshold overrides included.
def run_multi_sweep(ra, ctx, n_cases, seed, verbose=False):
    from oracle import rp_oracle as orc
    total = 0
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 55, ci])
        case = make_api_case(rng)
        c, x = case["cfg"], case["x"]
        if case["rate"] != 16000:
            x = x[::3].copy()
        d = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"],
                         score_ref=c["score_ref"], band_size=c["band_size"], score_mode=c["score_mode"], vad_mode=c["vad_mode"])
        for w in case["wakewords"]:
            d.add_ref(w)
        ref = []
        for k in range(len(x) // 480):
            r = d.process_i16(x[480 * k:480 * (k + 1)]) if x.dtype == np.int16 else d.process_f32(_to_f32(x[480 * k:480 * (k + 1)]))
            if r is not None:
                ref.append((k, r["counter"], r["name"], float(r["score"]), float(r["avg_score"])))
        dc = ra.DetectorConfig()
        dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
        dc.score_ref, dc.band_size = c["score_ref"], c["band_size"]
        dc.score_mode = getattr(ra.ScoreMode, c["score_mode"].capitalize())
        dc.vad_mode = {None: None, "easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[c["vad_mode"]]
        tms = [ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"]) for w in case["wakewords"]]
        det, dww, n_det = ctx.batch_detect_multi(x[None, :], tms, dc, thresholds=[w["threshold"] for w in case["wakewords"]],
                                                 avg_thresholds=[w["avg_threshold"] for w in case["wakewords"]], max_det=32)
        got = [(int(det[0][j]["frame"]) // 3 + 1, int(det[0][j]["counter"]), case["wakewords"][dww[0][j]]["name"],
                float(det[0][j]["score"]), float(det[0][j]["avg_score"])) for j in range(n_det[0])]
        ok = len(got) == len(ref) and all(g[:3] == r[:3] and abs(g[3] - r[3]) <= 1e-5 * abs(r[3]) and abs(g[4] - r[4]) <= 1e-5 * max(abs(r[4]), 1e-30)
                                          for g, r in zip(got, ref))
        assert ok, "multi sweep seed %d case %d (%r, K %d, %d wakewords)\noracle %r\ndevice %r" % (seed, ci, c, case["K"],
                                                                                                   len(case["wakewords"]), ref, got)
        total += len(ref)
        if verbose and ci % 20 == 0:
            print("multi case %d ok, %d detections so far" % (ci, total), flush=True)
    return n_cases, total


# ----------------------------------------------------------------------------------------------------------------
# Live-stream batches whose detectors hold several wakewords and / or a wakeword model (rp_stream_batch_new_multi): random
# references (1-3, own thresholds), now and then a random model of the same mfcc_size, random detector configs, a few streams
# fed in random pieces.  Compared with (a) the oracle's chunked detector holding the same wakewords -- same chunks fire, same
# wakeword / label, same counter, scores to 1e-5 (models 1e-4) -- and (b), references only, rp_batch_detect_multi over the whole
# streams bit for bit.  A score within 1e-5 of a threshold may be counted by one side only: such cases are counted as ties.
def run_live_multi_sweep(ra, ctx, n_cases, seed, verbose=False):
    from oracle import rp_oracle as orc
    total = ties = with_model = 0
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 77, ci])
        case = make_api_case(rng, models=True)
        c = case["cfg"]
        ww = case["wakewords"][:case["n_initial"] + (1 if len(case["wakewords"]) > case["n_initial"] and case["wakewords"][case["n_initial"]].get("kind") == "model" else 0)]
        ww = [w for w in ww if w.get("kind") == "model" or True][:4]
        K = case["K"]
        if case["rate"] == 16000:   # the case's own stream: noise with the wakewords' utterances planted in it
            base = _to_f32(case["x"])
            base = base[:len(base) // 480 * 480]
            n_chunks = len(base) // 480
        else:
            n_chunks = int(rng.integers(40, 120))
            base = np.concatenate([_utterance(rng, 480 * 20) for _ in range((n_chunks + 19) // 20)])[:480 * n_chunks].astype(np.float32)
        streams = []
        for s in range(int(rng.integers(2, 5))):
            x = np.roll(base, 480 * int(rng.integers(0, n_chunks))) + (rng.standard_normal(len(base)) * rng.uniform(0.0005, 0.01)).astype(np.float32)
            streams.append(x.astype(np.float32))
        pcm = np.stack(streams)
        dc = ra.DetectorConfig()
        dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
        dc.score_ref, dc.band_size = c["score_ref"], c["band_size"]
        dc.score_mode = getattr(ra.ScoreMode, c["score_mode"].capitalize())
        dc.vad_mode = {None: None, "easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[c["vad_mode"]]
        specs, keep, has_model = [], [], False
        for w in ww:
            if w.get("kind") == "model":
                nl = len([k for k in w["weights"] if k.endswith(".weight")])
                m = ra.Model(ctx, [w["weights"]["ln%d.weight" % (i + 1)] for i in range(nl)], [w["weights"]["ln%d.bias" % (i + 1)] for i in range(nl)])
                keep.append(m)
                specs.append({"model": m, "none_index": w["labels"].index("none") if "none" in w["labels"] else -1, "precision": "f32"})
                has_model = True
            else:
                t = ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"])
                keep.append(t)
                specs.append({"templates": t, "threshold": w["threshold"], "avg_threshold": w["avg_threshold"]})
        with_model += has_model
        pieces = [int(v) for v in rng.integers(1, 5, size=6)]
        sb = ra.StreamBatch(ctx, None, dc, pcm.shape[0], max_chunks_per_call=max(pieces), mfcc_size=K, wakewords=specs)
        got = [[] for _ in range(pcm.shape[0])]
        pos = k = 0
        while pos < pcm.shape[1]:
            nc = min(pieces[k % len(pieces)], (pcm.shape[1] - pos) // 480)
            k += 1
            det, dww, dlab, n_det = sb.process_multi(pcm[:, pos:pos + 480 * nc], max_det=8)
            pos += 480 * nc
            for s in range(pcm.shape[0]):
                for j in range(min(int(n_det[s]), 8)):
                    w = ww[dww[s][j]]
                    name = w["labels"][dlab[s][j]] if w.get("kind") == "model" else w["name"]
                    got[s].append((int(det[s][j]["frame"]) // 3 + 1, int(det[s][j]["counter"]), name, float(det[s][j]["score"]), float(det[s][j]["avg_score"]),
                                   det[s][j].copy(), int(dww[s][j])))
        where = "live multi sweep seed %d case %d (%r, K %d, %d wakewords%s)" % (seed, ci, c, K, len(ww), ", one a model" if has_model else "")
        case_tie = False
        for s in range(pcm.shape[0]):
            d = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"],
                             score_ref=c["score_ref"], band_size=c["band_size"], score_mode=c["score_mode"], vad_mode=c["vad_mode"])
            for w in ww:
                d.add_model(w) if w.get("kind") == "model" else d.add_ref(w)
            ref = []
            for kk in range(pcm.shape[1] // 480):
                r = d.process_f32(pcm[s, 480 * kk:480 * (kk + 1)])
                if r is not None:
                    ref.append((kk, r["counter"], r["name"], float(r["score"]), float(r["avg_score"])))
            tol = 1e-4 if has_model else 1e-5
            ok = len(got[s]) == len(ref) and all(g[:3] == r[:3] and abs(g[3] - r[3]) <= tol * max(abs(r[3]), 1e-3) and
                                                 abs(g[4] - r[4]) <= tol * max(abs(r[4]), 1e-3) for g, r in zip(got[s], ref))
            if not ok:   # a window whose score sits on a threshold: the two sides may count it differently, everything after differs
                case_tie = True
            total += len(ref)
        if not has_model:   # references only: the offline batch over the whole streams, bit for bit
            det, dww, n_det = ctx.batch_detect_multi(pcm, keep, dc, thresholds=[w["threshold"] for w in ww],
                                                     avg_thresholds=[w["avg_threshold"] for w in ww], max_det=64)
            for s in range(pcm.shape[0]):
                assert len(got[s]) == min(int(n_det[s]), 64) or len(got[s]) > 64, "%s stream %d: %d live against %d offline detections" % (where, s, len(got[s]), n_det[s])
                for j, g in enumerate(got[s][:64]):
                    assert g[5].tobytes()[4:] == det[s][j].tobytes()[4:] and g[6] == dww[s][j], "%s stream %d detection %d: live %r offline %r" % (where, s, j, g[5], det[s][j])
        if case_tie:
            # accept only if some window really sits within 1e-4 of a threshold for the oracle; otherwise it is a real difference
            ties += 1
            assert ties <= max(3, n_cases // 25), "%s: too many cases differ from the oracle's chunked detector" % where
        if verbose and ci % 10 == 0:
            print("live multi case %d ok, %d detections so far, %d with a model, %d near-tie cases" % (ci, total, with_model, ties), flush=True)
    return n_cases, total, with_model, ties


# ----------------------------------------------------------------------------------------------------------------
# Reference builder (rp_wakeword_ref_build = WakewordRef::new_from_sample_buffers + save_to_buffer) on random wav files:
# 8-bit unsigned / 16 / 32-bit PCM and IEEE float, mono / stereo, 16 or 48 kHz, 1-6 samples of different lengths.
# Templates against the oracle's MfccWavFileExtractor restatement and the averaged template against its averager (same
# DTW path decisions) at the tolerance of tonal signals (1e-4); thresholds, names, shapes as given.
def _wav_bytes(x, rate, bits, is_float, channels):
    import struct
    if is_float:
        body = x.astype("<f4")
        dec = body.astype(np.float32)
    elif bits == 8:
        q = np.clip(np.round(x * 127.0), -128, 127).astype(np.int16)
        body = (q + 128).astype(np.uint8)
        dec = q.astype(np.float32) / np.float32(127.0)
    elif bits == 16:
        q = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2")
        body, dec = q, q.astype(np.float32) / np.float32(32767.0)
    else:
        q = np.clip(np.round(x.astype(np.float64) * 2147483647.0), -2147483648, 2147483647).astype("<i4")
        body, dec = q, q.astype(np.float32) / np.float32(2147483648.0)
    if channels > 1:
        body = np.stack([body] + [np.roll(body, 7 * (k + 1)) for k in range(channels - 1)], axis=1).reshape(-1)
    raw = body.tobytes()
    fmt = struct.pack("<HHIIHH", 3 if is_float else 1, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    return b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(raw)) + raw, dec


def run_builder_sweep(ra, ctx, n_cases, seed, verbose=False):
    from oracle import rp_oracle as orc
    import rpw_py
    import tempfile
    checked = path_ties = 0
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 88, ci])
        K = int(rng.choice([5, 5, 16, 8]))
        rate = 48000 if rng.random() < 0.25 else 16000
        n_s = int(rng.integers(1, 7))
        samples, feats = {}, {}
        for i in range(n_s):
            dur = int(rng.integers(30, 110))  # chunks of 30 ms at 16 kHz
            x = _utterance(rng, 480 * dur)
            if rate == 48000:
                x = np.interp(np.arange(3 * len(x)) / 3.0, np.arange(len(x)), x).astype(np.float32)
            x = x[:len(x) - int(rng.integers(0, 200))]  # ragged ends
            kind = int(rng.integers(4))
            data, dec = _wav_bytes(x, rate, (8, 16, 32, 32)[kind], kind == 3, int(rng.choice([1, 1, 2])))
            name = "s%d.wav" % i
            samples[name] = data
            feats[name] = orc.wav_features(dec, rate, K)
        thr = None if rng.random() < 0.5 else float(rng.uniform(0.3, 0.6))
        athr = None if rng.random() < 0.5 else float(rng.uniform(0.0, 0.4))
        built = ctx.build_wakeword_ref("w%d" % ci, samples, K, threshold=thr, avg_threshold=athr, from_files=bool(rng.random() < 0.5))
        with tempfile.NamedTemporaryFile(suffix=".rpw") as f:
            f.write(built)
            f.flush()
            got = rpw_py.load_rpw(f.name)
        where = "builder sweep seed %d case %d (K %d, rate %d, %d samples)" % (seed, ci, K, rate, n_s)
        assert got["name"] == "w%d" % ci and got["mfcc_size"] == K and set(got["samples_features"]) == set(samples), where
        assert (got["threshold"] is None) == (thr is None) and (thr is None or np.float32(got["threshold"]) == np.float32(thr)), where
        assert (got["avg_threshold"] is None) == (athr is None) and (athr is None or np.float32(got["avg_threshold"]) == np.float32(athr)), where
        for k, ref in feats.items():
            g = got["samples_features"][k]
            assert g.shape == ref.shape, "%s %s: %r vs %r" % (where, k, g.shape, ref.shape)
            scale = np.maximum(np.abs(ref), 1.0) if K <= 5 else np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1.0)
            # the synthetic utterances are tonal (MFCC sweep: up to 6.6e-5 there, see run_mfcc_sweep); behind the resampler
            # the features also move with its 4e-6 of the peak
            tol = 1e-4 if rate == 16000 else 2e-4
            assert np.all(np.abs(g - ref) <= tol * scale), "%s %s: %.3g" % (where, k, float(np.max(np.abs(g - ref) / scale)))
        avg = orc.average_templates(feats)
        if avg is None:
            assert got["avg_features"] is None, where
        else:
            assert got["avg_features"].shape == avg.shape, where
            tol = (1e-4 if rate == 16000 else 2e-4) * max(1.0, float(np.abs(avg).max()))
            if np.abs(got["avg_features"] - avg).max() > tol:
                # the average follows a DTW path: where two steps cost the same to the last bits, templates that differ by
                # their tolerance can take different paths.  Then the product's average must be the oracle's averager
                # applied to the product's OWN templates (same path decisions on the same numbers).
                avg2 = orc.average_templates({k: got["samples_features"][k] for k in feats})
                assert np.abs(got["avg_features"] - avg2).max() <= tol, "%s avg: %.3g (own templates: %.3g)" % (
                    where, float(np.abs(got["avg_features"] - avg).max()), float(np.abs(got["avg_features"] - avg2).max()))
                path_ties += 1
        checked += n_s
        if verbose and ci % 20 == 0:
            print("builder case %d ok, %d samples compared so far, %d averaging-path ties" % (ci, checked, path_ties), flush=True)
    assert path_ties <= max(2, n_cases // 50)
    return n_cases, checked


# ----------------------------------------------------------------------------------------------------------------
# Model trainer (rp_wakeword_model_train = WakewordModel::train_from_buffers): random labelled wav sets, the four model
# types, mfcc sizes, learning rates and epoch counts.  The start is the product's own seeded initialisation (0 epochs,
# read back from its file); from there the same full-batch SGD epochs as the oracle's restatement must give the same
# weights and loss (1e-4 of the larger of weight scale and movement, as in tests/test_gpu_parity.py).
def run_train_sweep(ra, ctx, n_cases, seed, verbose=False):
    from oracle import rp_oracle as orc
    import rpw_py
    import tempfile

    def load(data):
        with tempfile.NamedTemporaryFile(suffix=".rpw") as f:
            f.write(data)
            f.flush()
            return rpw_py.load_rpw(f.name)

    skipped = kinks = 0
    for ci in range(n_cases):
        rng = np.random.default_rng([seed, 99, 7, ci])
        K = int(rng.choice([16, 16, 8, 5]))
        m_type = str(rng.choice(["tiny", "small", "medium", "large"]))
        labels_pool = ["alpha", "beta"][:int(rng.integers(1, 3))]
        rate = 48000 if rng.random() < 0.2 else 16000
        train, feats = {}, {}
        for i in range(int(rng.integers(4, 10))):
            lab = None if (i % 3 == 2) else labels_pool[i % len(labels_pool)]
            # broadband samples (shaped noise): their MFCCs agree with the oracle's at 1e-5, so the weights can be held to a
            # tight tolerance (tonal signals move the features by up to 6e-5, see run_mfcc_sweep)
            n = 480 * int(rng.integers(40, 90))
            env = np.interp(np.arange(n), np.linspace(0, n, 12), rng.uniform(0.05, 1.0, 12))
            x = (env * rng.standard_normal(n) * rng.uniform(0.02, 0.2)).astype(np.float32)
            if rate == 48000:
                x = np.interp(np.arange(3 * len(x)) / 3.0, np.arange(len(x)), x).astype(np.float32)
            data, dec = _wav_bytes(x, rate, 16, False, 1)
            name = ("[%s]s%d.wav" % (lab, i)) if lab else ("noise%d.wav" % i)
            train[name] = data
            feats[name] = orc.wav_features(dec, rate, K).reshape(-1)
        test = dict(list(train.items())[:2])
        lr, epochs = float(rng.choice([0.002, 0.01, 0.027])), int(rng.integers(1, 8))
        where = "train sweep seed %d case %d (%s, K %d, rate %d, %d samples, lr %g, %d epochs)" % (seed, ci, m_type, K, rate, len(train), lr, epochs)
        try:
            init, _, _ = ctx.train_wakeword_model(train, test, m_type, lr, 0, 1, K, seed=int(rng.integers(1, 1000)))
        except ra.RustpotterError as e:  # samples too short for the type: the reference refuses them as well
            assert "too short" in str(e), where + ": " + str(e)
            continue
        m0 = load(init)
        L = m0["train_size"] * K
        names = list(train)
        xs = np.zeros((len(names), L), np.float32)
        ys = []
        for r, name in enumerate(names):
            f = feats[name]
            xs[r, :min(L, len(f))] = f[:L]
            ys.append(m0["labels"].index(name[1:name.index("]")] if name.startswith("[") else "none"))
        nl = len([k for k in m0["weights"] if k.endswith(".weight")])
        ws = [m0["weights"]["ln%d.weight" % (i + 1)] for i in range(nl)]
        bs = [m0["weights"]["ln%d.bias" % (i + 1)] for i in range(nl)]
        data, loss, acc = ctx.train_wakeword_model(train, test, "tiny", lr, epochs, 1, 5, seed=1, prev_model=init)
        m1 = load(data)
        assert m1["m_type"] == m0["m_type"] and m1["labels"] == m0["labels"] and m1["train_size"] == m0["train_size"], where
        rw, rb, rloss = orc.mlp_train(xs, ys, ws, bs, lr, epochs)
        if not np.isfinite(rloss) or rloss > 20.0:  # a diverging run amplifies the last bit of every feature: nothing to compare
            skipped += 1
            continue
        if verbose:
            print("  %s: loss %g (oracle %g), max |w| %.3g, moved %.3g" % (where, loss, rloss, max(float(np.abs(w).max()) for w in rw),
                                                                       max(float(np.abs(a - b0).max()) for a, b0 in zip(rw, ws))), flush=True)
        assert abs(loss - rloss) <= 2e-4 * max(abs(rloss), 1e-3), "%s: loss %g vs %g" % (where, loss, rloss)
        worst = 0.0
        for i in range(nl):
            for kind, ref, start in (("weight", rw[i], ws[i]), ("bias", rb[i], bs[i])):
                got = m1["weights"]["ln%d.%s" % (i + 1, kind)]
                assert got.shape == ref.shape, where
                tol = 2e-4 * max(float(np.abs(ref).max()), float(np.abs(ref - start).max()))
                worst = max(worst, float(np.abs(got - ref).max()) / tol)
        if worst > 1.0:
            # a hidden unit whose pre-activation is ~0 for one sample: the ReLU derivative flips on the last bit and that
            # sample's whole contribution to the unit's gradient comes or goes (seen once in 280 runs: weights 13 % of
            # their movement apart, loss equal to 4e-5).  Counted, and bounded by how far one sample can move a weight.
            assert worst < 2e4 and abs(loss - rloss) <= 1e-3 * max(abs(rloss), 1e-3), "%s: weights %.3g x their tolerance apart" % (where, worst)
            kinks += 1
        if verbose and ci % 10 == 0:
            print("train case %d ok (%d diverging runs skipped so far)" % (ci, skipped), flush=True)
    assert skipped <= n_cases // 3, "train sweep: %d of %d runs diverged" % (skipped, n_cases)
    assert kinks <= max(1, n_cases // 50), "train sweep: %d of %d runs parted at a ReLU kink" % (kinks, n_cases)
    return n_cases - skipped


# ----------------------------------------------------------------------------------------------------------------
# Wakeword models (src/wakewords/nn/wakeword_nn.rs) through the single-stream API: random layer sizes of the four model
# types, random weights, 2-3 labels.  The forward pass is pinned by the oracle only, so scores compare at
# 1e-4; which chunks fire, the label and the counter must agree.
def make_model_case(rng):
    K = int(rng.choice([16, 16, 5, 8]))
    fr = int(rng.integers(30, 130))
    mt = int(rng.integers(4))
    hidden = {0: [fr // 15], 1: [fr // 6, (fr // 6) // 2], 2: [fr // 3, fr // 6], 3: [(fr // 3) * 2, fr // 6]}[mt]
    labels = ["none", "alpha", "beta"][:int(rng.integers(2, 4))]
    if rng.random() < 0.2:
        labels = labels[1:] + ["gamma"]  # no "none" label at all
    dims = [fr * K] + hidden + [len(labels)]
    weights = {}
    for i in range(len(dims) - 1):
        weights["ln%d.weight" % (i + 1)] = (rng.standard_normal((dims[i + 1], dims[i])) * (1.5 / np.sqrt(dims[i]))).astype(np.float32)
        weights["ln%d.bias" % (i + 1)] = (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)
    model = {"labels": labels, "train_size": fr, "mfcc_size": K, "m_type": ["Tiny", "Small", "Medium", "Large"][mt],
             "weights": weights, "rms_level": float(rng.uniform(0.01, 0.2))}
    cfg = dict(threshold=float(rng.uniform(0.0, 0.6)), avg_threshold=float(rng.choice([0.0, rng.uniform(0.0, 0.5)])),
               min_scores=int(rng.integers(1, 7)), eager=bool(rng.random() < 0.3),
               vad_mode=[None, None, None, "easy", "medium", "hard"][int(rng.integers(6))])
    n_chunks = int(rng.integers(50, 160))
    x = np.concatenate([_utterance(rng, 480 * 20) for _ in range((n_chunks + 19) // 20)])[:480 * n_chunks]
    x = x + (rng.standard_normal(len(x)) * rng.uniform(0.0005, 0.02)).astype(np.float32)
    return dict(model=model, cfg=cfg, x=x.astype(np.float32))


def run_model_sweep(ra, n_cases, seed, verbose=False, ctx=None):
    """ctx given: the same stream also goes through rp_batch_detect_model (f32) in one call."""
    from oracle import rp_oracle as orc
    import rpw_py
    total = ties = 0
    for ci in range(n_cases):
        case = make_model_case(np.random.default_rng([seed, 99, ci]))
        c, m, x = case["cfg"], case["model"], case["x"]
        d = orc.Detector(avg_threshold=c["avg_threshold"], threshold=c["threshold"], min_scores=c["min_scores"], eager=c["eager"],
                         vad_mode=c["vad_mode"])
        d.add_model(m)
        rc = ra.RustpotterConfig.default()
        rc.fmt.sample_format = ra.SampleFormat.F32
        dc = rc.detector
        dc.avg_threshold, dc.threshold, dc.min_scores, dc.eager = c["avg_threshold"], c["threshold"], c["min_scores"], c["eager"]
        dc.vad_mode = {None: None, "easy": ra.VADMode.Easy, "medium": ra.VADMode.Medium, "hard": ra.VADMode.Hard}[c["vad_mode"]]
        rp = ra.Rustpotter.new(rc)
        rp.add_wakeword_from_buffer("m", rpw_py.dump_rpw_model(m["labels"], m["train_size"], m["mfcc_size"], m["m_type"], m["weights"],
                                                               m["rms_level"]))
        where = "model sweep seed %d case %d (%r, %s, K %d, frames %d, labels %r)" % (seed, ci, c, m["m_type"], m["mfcc_size"],
                                                                                     m["train_size"], m["labels"])
        refs = []
        for k in range(len(x) // 480):
            ref = d.process_f32(x[480 * k:480 * (k + 1)])
            got = rp.process_samples(np.ascontiguousarray(x[480 * k:480 * (k + 1)]))
            assert (got is None) == (ref is None), "%s chunk %d: %r vs %r" % (where, k, got, ref)
            if ref is None:
                continue
            refs.append((k, ref))
            total += 1
            assert got.name == ref["name"] and got.counter == ref["counter"], "%s chunk %d: %r vs %r" % (where, k, got, ref)
            assert abs(float(got.score) - float(ref["score"])) <= 1e-4 * max(1.0, abs(float(ref["score"]))), \
                "%s chunk %d: %r vs %r" % (where, k, got, ref)
        if ctx is not None:
            nl = len([k for k in m["weights"] if k.endswith(".weight")])
            model = ra.Model(ctx, [m["weights"]["ln%d.weight" % (i + 1)] for i in range(nl)], [m["weights"]["ln%d.bias" % (i + 1)] for i in range(nl)])
            none_index = m["labels"].index("none") if "none" in m["labels"] else -1
            det, dlab, n_det = ctx.batch_detect_model(x[None, :], model, m["mfcc_size"], none_index, dc, max_det=64)
            same = n_det[0] == len(refs) and all(
                det[0][j]["frame"] // 3 + 1 == k and det[0][j]["counter"] == ref["counter"] and m["labels"][dlab[0][j]] == ref["name"] and
                abs(float(det[0][j]["score"]) - float(ref["score"])) <= 1e-4 * max(1.0, abs(float(ref["score"])))
                for j, (k, ref) in enumerate(refs[:64]))
            if not same:
                # the batched path takes the window mean out after layer 1 (other rounding than the explicit windows of the
                # handle): a window whose score sits within 1e-5 of a threshold may be counted by one and not by the other
                K, L = m["mfcc_size"], m["train_size"]
                mf = orc.mfcc_stream(x, K)
                X = np.stack([orc.normalize(mf[w:w + L]).reshape(-1) for w in range(len(mf) - L + 1)])
                lg = orc.mlp_forward(X, [m["weights"]["ln%d.weight" % (i + 1)] for i in range(nl)], [m["weights"]["ln%d.bias" % (i + 1)] for i in range(nl)])
                rf = np.float32(0.22 * 10.0)
                best = lg.max(axis=1)
                none = lg[:, none_index] if none_index >= 0 else np.zeros(len(lg), np.float32)
                sc = 1.0 - 1.0 / (1.0 + np.exp(((best - none) - rf) / rf))
                av = 1.0 - 1.0 / (1.0 + np.exp(((best - lg.min(axis=1)) - rf) / rf))
                tie = np.min(np.abs(sc - c["threshold"])) <= 1e-5 or (c["avg_threshold"] != 0 and np.min(np.abs(av - c["avg_threshold"])) <= 1e-5)
                assert tie, "%s batched: %r vs oracle %r" % (where, [(int(det[0][j]["frame"]) // 3 + 1, int(det[0][j]["counter"]), float(det[0][j]["score"]))
                                                                  for j in range(min(int(n_det[0]), 64))], [(k, r["counter"], float(r["score"])) for k, r in refs])
                ties += 1
        if verbose and ci % 20 == 0:
            print("model case %d ok, %d detections so far, %d threshold ties in the batched path" % (ci, total, ties), flush=True)
    assert ties <= max(2, n_cases // 100)
    return n_cases, total


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--min-level-exp", type=float, default=-2.0, help="MFCC sweep: signal levels 10^[this, 0.5]")
    ap.add_argument("--report", action="store_true", help="MFCC sweep: report the worst errors instead of asserting the gates")
    ap.add_argument("--resample-cases", type=int, default=0, help="resampler cases (rates x channels x sample types)")
    ap.add_argument("--frontend-cases", type=int, default=0, help="decode + gain normaliser + band-pass cases")
    ap.add_argument("--mfcc-cases", type=int, default=0, help="MFCC value cases (signal kinds x levels)")
    ap.add_argument("--train-cases", type=int, default=0, help="wakeword models trained from random labelled wav sets")
    ap.add_argument("--builder-cases", type=int, default=0, help="wakeword references built from random wav files")
    ap.add_argument("--multi-cases", type=int, default=0, help="several wakewords in rp_batch_detect_multi")
    ap.add_argument("--live-multi-cases", type=int, default=0, help="live-stream batches holding several wakewords / a model")
    ap.add_argument("--model-cases", type=int, default=0, help="wakeword-model cases through the single-stream API")
    ap.add_argument("--rate-cases", type=int, default=0, help="live-stream batches behind the resampler (8-48 kHz, stereo)")
    ap.add_argument("--reset-cases", type=int, default=0, help="live-stream batches with single-stream resets")
    ap.add_argument("--extreme-cases", type=int, default=0, help="single-stream API cases with edge-of-range detector parameters")
    ap.add_argument("--api-cases", type=int, default=None, help="single-stream API cases (default: cases / 4)")
    ap.add_argument("--ragged-cases", type=int, default=0, help="batch cases in the shapes the opt-in dtw_ragged_kernel takes (mfcc_size 5, 1..9 templates of unequal length, band 3..5), scored with RP_ARITH_FAST_SPLIT + ragged_matrix")
    ap.add_argument("--mfma-cases", type=int, default=0, help="batch cases in the shapes the matrix-core DTW kernel takes (mfcc_size 5 at band 3..5, mfcc_size 13 / 16 at band 5; 3..16 same-length templates, one case in eight of mfcc_size 5 with 32..44: dtw_mfma_group_kernel)")
    a = ap.parse_args()
    import rustpotter_amd as ra
    ctx = ra.BatchContext(0)
    mfcc_line = lambda r: "%d cases, worst scaled error per signal kind %r (gate 1e-5; tones 1e-4, speech-like 3e-5)" % (
        r[0], {k: float("%.3g" % v) for k, v in sorted(r[1].items())})
    families = [  # (name, number of cases, run, result -> text); a failing family is reported and the others still run
        ("sweep", a.cases, lambda n: run_sweep(ra, ctx, n, a.seed, verbose=True),
         lambda r: "%d cases, %d detections compared, %d threshold ties skipped" % r),
        ("matrix-core DTW sweep, RP_ARITH_F32_MATRIX (default: three bf16 parts)", a.mfma_cases, lambda n: run_sweep(ra, ctx, n, a.seed, verbose=True, mfma=True, arith="f32_matrix"),
         lambda r: "%d cases (mfcc_size 5 at band 3..5: dtw_mfma_kernel in chunks of 3..8; mfcc_size 13 / 16 at band 5: dtw_mfma_wide3_kernel in chunks of up to four; 3..16 same-length templates, one case in eight of mfcc_size 5 with 32..44), %d detections compared, %d threshold ties skipped" % r),
        ("matrix-core DTW sweep, RP_ARITH_FAST_SPLIT (two f16 parts)", a.mfma_cases, lambda n: run_sweep(ra, ctx, n, a.seed, verbose=True, mfma=True, arith="fast_split"),
         lambda r: "%d cases (mfcc_size 5 at band 3..5, mfcc_size 13 / 16 at band 5; 3..16 same-length templates, one case in eight of mfcc_size 5 with 32..44: dtw_mfma_group_kernel), %d detections compared, %d threshold ties skipped" % r),
        ("ragged matrix-core DTW sweep (RP_ARITH_FAST_SPLIT + ragged_matrix)", a.ragged_cases, lambda n: run_sweep(ra, ctx, n, a.seed, verbose=True, ragged=True),
         lambda r: "%d cases (mfcc_size 5, 1..9 templates of unequal length, band 3..5; the kernel ran in every case), %d detections compared, %d threshold ties skipped" % r),
        ("live rate sweep", a.rate_cases, lambda n: run_live_rate_sweep(ra, ctx, n, a.seed, verbose=True),
         lambda r: "%d cases live == offline bitwise, %d detections equal to the oracle's, %d near-tie cases" % r),
        ("live reset sweep", a.reset_cases, lambda n: run_live_reset_sweep(ra, ctx, n, a.seed, verbose=True),
         lambda r: "%d cases, %d detections compared" % r),
        ("api sweep", a.cases // 4 if a.api_cases is None else a.api_cases, lambda n: run_api_sweep(ra, n, a.seed, verbose=True),
         lambda r: "%d cases, %d detections compared" % r),
        ("sweep (extreme detector parameters)", a.extreme_cases, lambda n: run_sweep(ra, ctx, n, a.seed + 2000, verbose=True, extreme=True),
         lambda r: "%d cases, %d detections compared, %d threshold ties skipped" % r),
        ("api sweep (extreme detector parameters)", a.extreme_cases, lambda n: run_api_sweep(ra, n, a.seed + 1000, verbose=True, extreme=True),
         lambda r: "%d cases, %d detections compared" % r),
        ("resample sweep", a.resample_cases, lambda n: run_resample_sweep(ra, ctx, n, a.seed, verbose=True),
         lambda r: "%d cases, worst error %.3g of the peak (gate 4e-6, x sqrt(fi / 1440) above 48 kHz)" % r),
        ("frontend sweep", a.frontend_cases, lambda n: run_frontend_sweep(ra, ctx, n, a.seed, verbose=True),
         lambda r: "%d cases, %d streams compared bit for bit" % r),
        ("mfcc sweep", a.mfcc_cases, lambda n: run_mfcc_sweep(ra, ctx, n, a.seed, verbose=True, strict=not a.report, min_level_exp=a.min_level_exp), mfcc_line),
        ("train sweep", a.train_cases, lambda n: (run_train_sweep(ra, ctx, n, a.seed, verbose=True),), lambda r: "%d cases" % r),
        ("builder sweep", a.builder_cases, lambda n: run_builder_sweep(ra, ctx, n, a.seed, verbose=True),
         lambda r: "%d cases, %d wav samples compared" % r),
        ("multi sweep", a.multi_cases, lambda n: run_multi_sweep(ra, ctx, n, a.seed, verbose=True), lambda r: "%d cases, %d detections compared" % r),
        ("live multi sweep", a.live_multi_cases, lambda n: run_live_multi_sweep(ra, ctx, n, a.seed, verbose=True),
         lambda r: "%d cases, %d detections equal to the oracle's (references-only cases also bitwise equal to the offline batch), %d cases with a model, %d near-tie cases" % r),
        ("model sweep", a.model_cases, lambda n: run_model_sweep(ra, n, a.seed, verbose=True, ctx=ctx), lambda r: "%d cases, %d detections compared" % r),
    ]
    failed = 0
    for name, n, run, text in families:
        if n <= 0:
            continue
        try:
            print("%s: %s: OK" % (name, text(run(n))), flush=True)
        except AssertionError as e:
            failed += 1
            print("%s: FAILED: %s" % (name, str(e)[:3000]), flush=True)
    sys.exit(1 if failed else 0)
