"""GPU tests added in round 2 (all through the C ABI): the averaged-template gate as a skip, the multi-GPU paths
(two ranks sharing one GPU over gloo, rp_batch_detect_sharded, bench.py starting its own ranks), BASELINE configs C2 and
C5 at their full sizes, and NULL-argument handling with live handles."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import rpw_py
import simstream
from oracle import rp_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = simstream.GOLDEN
EXP = json.load(open(os.path.join(G, "expectations.json")))
SEED = 0x5EED000000000001


@pytest.fixture(scope="module")
def ra():
    import rustpotter_amd
    return rustpotter_amd


def _fixture_streams(n_variants=6):
    """Variants of the reference's simulation stream (tests/detector.rs:372-426): shifted by whole chunks, with noise."""
    base = simstream.i16_to_f32(simstream.simulation_stream_i16())
    rng = np.random.default_rng(7)
    n = (len(base) // 480) * 480
    out = [base[:n]]
    for i in range(1, n_variants):
        v = np.roll(base, 480 * (3 + 5 * i))
        if i % 2:
            v = v + rng.standard_normal(len(base)).astype(np.float32) * np.float32(0.001 * i)
        out.append(v[:n].astype(np.float32))
    return np.stack(out)


def _devices(n):
    """Device ordinals for n shards / ranks: DISTINCT devices the moment the node has that many (then the peer-copy
    gather crosses xGMI and the ranks talk RCCL); on a smaller node the shards share what there is."""
    import torch
    have = max(1, torch.cuda.device_count())
    return [g % have for g in range(n)]


def _distinct(n):
    import torch
    return torch.cuda.device_count() >= n


def _wakeword(ra, ctx, name="oye_casa_g.rpw"):
    w = rpw_py.load_rpw(os.path.join(G, name))
    return ra.Templates(ctx, list(w["samples_features"].values()), avg=w["avg_features"])


# ------------------------------------------------------------------ the averaged-template gate as a skip
@pytest.mark.parametrize("avg_threshold,threshold", [(0.2, 0.5), (0.4, 0.45), (0.55, 0.45), (0.62, 0.3), (0.9, 0.3)])
def test_avg_gate_skip_gives_the_detections_of_full_scoring(ra, avg_threshold, threshold):
    """wakeword_comp.rs:85-93: a window whose avg_score is below avg_threshold is not compared with the sample templates.
    The skipping path (one DTW for gated windows) and RP_CTX_FULL_SCORES (T+1 DTWs for every window) must report the
    same detections, field by field and bit by bit, whatever part of the windows the gate removes."""
    pcm = _fixture_streams()
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    cfg = ra.DetectorConfig()
    cfg.avg_threshold, cfg.threshold = avg_threshold, threshold
    det_g, n_g = gated.batch_detect(pcm, _wakeword(ra, gated), cfg)
    det_f, n_f = full.batch_detect(pcm, _wakeword(ra, full), cfg)
    assert np.array_equal(n_g, n_f)
    assert det_g.tobytes() == det_f.tobytes()
    # the per-window arrays stay complete when they are asked for (then nothing is skipped)
    det_s, n_s, scores, agg = gated.batch_detect(pcm, _wakeword(ra, gated), cfg, want_scores=True)
    assert np.array_equal(n_s, n_f) and det_s.tobytes() == det_f.tobytes()
    assert np.isfinite(scores).all() and np.isfinite(agg).all()
    if avg_threshold <= 0.55:
        assert n_f.sum() >= 1   # the planted utterances are found
    if avg_threshold >= 0.9:
        assert n_f.sum() == 0   # nothing passes the gate: the list is empty and every tile of pass 3 exits


def test_avg_gate_skip_on_synthetic_noise_many_streams(ra):
    """Many streams whose windows straddle the gate (threshold at the median avg_score): detections with a low score
    threshold must agree between the two paths; band sizes 3..6 take the same route."""
    S, N, K, L, T = 700, 480 * 60, 5, 40, 5
    templates = orc.synth_templates(SEED, T, L, K)
    avg = np.mean(templates, axis=0, dtype=np.float32)
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    pcm = gated.synth_pcm(SEED, 0, S, N)
    tg, tf = ra.Templates(gated, templates, avg=avg), ra.Templates(full, templates, avg=avg)
    mf = gated.mfcc(pcm, K)
    for band, q in ((5, 0.5), (3, 0.5), (6, 0.5), (5, 0.0), (5, 0.03), (4, 0.97)):
        # q: the quantile of the avg scores the gate sits at -- 0.5: half of the windows are listed (list mode); 0 / 0.03: all /
        # nearly all pass (the dense form of pass 3: every window through the staged kernels); 0.97: a short list
        _, av, ag = gated.dtw_scores(mf, tg, band_size=band, with_avg=True)
        cfg = ra.DetectorConfig()
        cfg.band_size = band
        cfg.avg_threshold = float(np.quantile(av, q))
        cfg.threshold = float(np.quantile(ag, 0.7))
        cfg.min_scores = 2
        det_g, n_g = gated.batch_detect(pcm, tg, cfg, max_det=6)
        det_f, n_f = full.batch_detect(pcm, tf, cfg, max_det=6)
        assert n_f.sum() > (S // 4 if q <= 0.5 else 0), "the case must produce detections"
        assert np.array_equal(n_g, n_f) and det_g.tobytes() == det_f.tobytes(), (band, q)


@pytest.mark.parametrize("K", [13, 16])
def test_avg_gate_skip_wide_frames(ra, K):
    """The same for mfcc_size 13 / 16 (dtw_band_wide_kernel in list mode), ragged templates so that one- and two-template
    chunks both occur."""
    S, N, L, T = 300, 480 * 50, 40, 5
    templates = orc.synth_templates(SEED, T, L, K)
    templates[1] = templates[1][:L - 5].copy()
    templates[3] = templates[3][:L - 9].copy()
    avg = np.mean([t[:L - 9] for t in templates], axis=0, dtype=np.float32)
    gated, full = ra.BatchContext(0), ra.BatchContext(0, full_scores=True)
    pcm = gated.synth_pcm(SEED, 0, S, N)
    tg, tf = ra.Templates(gated, templates, avg=avg), ra.Templates(full, templates, avg=avg)
    mf = gated.mfcc(pcm, K)
    for band in (5, 4):
        _, av, ag = gated.dtw_scores(mf, tg, band_size=band, with_avg=True)
        cfg = ra.DetectorConfig()
        cfg.band_size, cfg.min_scores = band, 2
        cfg.avg_threshold, cfg.threshold = float(np.median(av)), float(np.quantile(ag, 0.7))
        det_g, n_g = gated.batch_detect(pcm, tg, cfg, max_det=6)
        det_f, n_f = full.batch_detect(pcm, tf, cfg, max_det=6)
        assert n_f.sum() > S // 4 and np.array_equal(n_g, n_f) and det_g.tobytes() == det_f.tobytes()


@pytest.mark.parametrize("avg_threshold,chunks_per_call", [(0.2, 1), (0.5, 1), (0.5, 4), (0.62, 7)])
def test_avg_gate_skip_in_live_stream_batches(ra, avg_threshold, chunks_per_call):
    """rp_stream_batch_process skips the sample templates of gated windows too: fed chunk by chunk it reports the detections
    of the offline call that scores everything."""
    pcm = _fixture_streams(5)
    cfg = ra.DetectorConfig()
    cfg.avg_threshold, cfg.threshold = avg_threshold, 0.45
    full = ra.BatchContext(0, full_scores=True)
    det_f, n_f = full.batch_detect(pcm, _wakeword(ra, full), cfg, max_det=6)
    ctx = ra.BatchContext(0)
    sb = ra.StreamBatch(ctx, _wakeword(ra, ctx), cfg, pcm.shape[0], max_chunks_per_call=chunks_per_call)
    live = [[] for _ in range(pcm.shape[0])]
    step = 480 * chunks_per_call
    for i in range(0, pcm.shape[1], step):
        d, nd = sb.process(np.ascontiguousarray(pcm[:, i:i + step]), max_det=8)
        for s in range(pcm.shape[0]):
            live[s] += [d[s][j] for j in range(nd[s])]
    for s in range(pcm.shape[0]):
        assert len(live[s]) == n_f[s]
        for a, b in zip(live[s], det_f[s][:n_f[s]]):
            assert (a["frame"], a["window"], a["counter"]) == (b["frame"], b["window"], b["counter"])
            assert a["score"] == b["score"] and a["avg_score"] == b["avg_score"]
    assert n_f.sum() >= (5 if avg_threshold <= 0.5 else 0)


def test_avg_gate_skip_with_several_wakewords(ra):
    """rp_batch_detect_multi: each wakeword's own avg_threshold gates its own sample templates; same detections and same
    firing wakeword as the path that scores everything."""
    pcm = _fixture_streams(4)
    cfg = ra.DetectorConfig()
    cfg.threshold, cfg.min_scores = 0.45, 3
    outs = []
    for full in (False, True):
        ctx = ra.BatchContext(0, full_scores=full)
        tms = [_wakeword(ra, ctx, "oye_casa_g.rpw"), _wakeword(ra, ctx, "alexa.rpw")]
        outs.append(ctx.batch_detect_multi(pcm, tms, cfg, avg_thresholds=[0.5, 0.3]))
    (d0, w0, n0), (d1, w1, n1) = outs
    assert np.array_equal(n0, n1) and n0.sum() >= 4 and d0.tobytes() == d1.tobytes() and np.array_equal(w0, w1)


# ------------------------------------------------------------------ early abandon in detect-only calls
@pytest.mark.parametrize("K,lens", [(5, [40] * 8), (5, [40, 36, 33, 40, 31]), (13, [40, 40, 35]), (16, [38, 40])])
def test_early_abandon_never_changes_a_detection(ra, K, lens):
    """Detect-only calls in ScoreMode::Max stop DTWs that can no longer reach `threshold` (launch_dtw, abandon_nc).  With
    the threshold placed INSIDE the score distribution (many windows just above and just below it) the detections must
    equal, bit for bit, those of the call that returns every score -- offline and fed chunk by chunk."""
    S, N = 400, 480 * 60
    templates = [t[:n].copy() for t, n in zip(orc.synth_templates(SEED, len(lens), max(lens), K), lens)]
    ctx = ra.BatchContext(0)
    pcm = ctx.synth_pcm(SEED, 0, S, N)
    tm = ra.Templates(ctx, templates)
    for q, min_scores in ((0.5, 1), (0.9, 2), (0.99, 1), (0.999, 1)):
        cfg = ra.DetectorConfig()
        cfg.avg_threshold, cfg.min_scores = 0.0, min_scores
        d0, n0, scores, agg = ctx.batch_detect(pcm, tm, cfg, max_det=8, want_scores=True)   # every score, nothing abandoned
        cfg.threshold = float(np.quantile(agg, q))
        d_full, n_full, _, _ = ctx.batch_detect(pcm, tm, cfg, max_det=8, want_scores=True)
        d_fast, n_fast = ctx.batch_detect(pcm, tm, cfg, max_det=8)                            # detect-only: may abandon
        assert n_full.sum() > 20, "the case must produce detections"
        assert np.array_equal(n_fast, n_full) and d_fast.tobytes() == d_full.tobytes(), (K, q)
    # live-stream batches take the same shortcut when the per-window aggregates are not asked for
    sb = ra.StreamBatch(ctx, tm, cfg, S, max_chunks_per_call=3)
    live = [[] for _ in range(S)]
    for i in range(0, N, 480 * 3):
        d, nd = sb.process(np.ascontiguousarray(pcm[:, i:i + 480 * 3]), max_det=8)
        for s_ in range(S):
            live[s_] += [d[s_][j] for j in range(nd[s_])]
    for s_ in range(S):
        assert len(live[s_]) == n_full[s_]
        for a, b in zip(live[s_], d_full[s_][:min(n_full[s_], 8)]):
            assert (a["frame"], a["window"], a["counter"], a["score"]) == (b["frame"], b["window"], b["counter"], b["score"])


def test_early_abandon_degenerate_thresholds(ra):
    """threshold <= 0 (everything fires), >= 1 (nothing can), and a threshold so high that every DTW is abandoned at the
    first check: same detections as the full path."""
    S, N, K = 64, 480 * 40, 5
    templates = orc.synth_templates(SEED, 3, 30, K)
    ctx = ra.BatchContext(0)
    pcm = ctx.synth_pcm(SEED, 0, S, N)
    tm = ra.Templates(ctx, templates)
    for thr in (-1.0, 0.0, 0.999, 1.0, 1.5):
        cfg = ra.DetectorConfig()
        cfg.avg_threshold, cfg.threshold, cfg.min_scores = 0.0, thr, 1
        d_full, n_full, _, _ = ctx.batch_detect(pcm, tm, cfg, max_det=4, want_scores=True)
        d_fast, n_fast = ctx.batch_detect(pcm, tm, cfg, max_det=4)
        assert np.array_equal(n_fast, n_full) and d_fast.tobytes() == d_full.tobytes(), thr
    # (with every window above the threshold the countdown never reaches zero: eager mode makes those cases fire)
    cfg.threshold, cfg.eager = 0.0, True
    d_full, n_full, _, _ = ctx.batch_detect(pcm, tm, cfg, max_det=4, want_scores=True)
    d_fast, n_fast = ctx.batch_detect(pcm, tm, cfg, max_det=4)
    assert n_full.sum() >= S and np.array_equal(n_fast, n_full) and d_fast.tobytes() == d_full.tobytes()


# ------------------------------------------------------------------ multi-GPU: ranks, shards, gather
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, out_path, backend, devices):
    """One rank of the data-parallel path as bench.py runs it: its shard of the streams through rp_batch_detect on its
    device, then ONE all_gather of the per-stream results -- RCCL (backend nccl) with one device per rank when the node
    has them, gloo with the ranks sharing GPU 0 otherwise."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import rustpotter_amd as ra
    from rustpotter_amd import sharding
    dev_id = devices[rank]
    torch.cuda.set_device(dev_id)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_id))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pcm_all = _fixture_streams(8)
        lo, hi = sharding.shard_bounds(pcm_all.shape[0], world, rank)
        ctx = ra.BatchContext(device=dev_id, host_pointers=False)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        tm = _wakeword(ra, ctx)
        cfg = ra.DetectorConfig()
        cfg.threshold = 0.45
        S, N = hi - lo, pcm_all.shape[1]
        nf = ra.mfcc_num_frames(N)
        n_win = nf - tm.max_len + 1
        pcm = torch.from_numpy(pcm_all[lo:hi].copy()).cuda()
        det = torch.zeros((S, 4, 6), dtype=torch.int32, device="cuda")
        n_det = torch.zeros((S,), dtype=torch.int32, device="cuda")
        scores = torch.empty((S, n_win, tm.T), dtype=torch.float32, device="cuda")
        agg = torch.empty((S, n_win), dtype=torch.float32, device="cuda")
        ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tm, cfg, det.data_ptr(), n_det.data_ptr(), 4, scores.data_ptr(), agg.data_ptr())
        torch.cuda.synchronize()
        used = torch.arange(4, device="cuda")[None, :] < n_det[:, None]
        det[:, :, 0] += lo * used.to(torch.int32)  # global stream ids in the slots that hold a detection (the others stay zero)
        chk = scores.view(torch.int32).to(torch.int64).sum(dim=(1, 2))  # per-stream checksum of the score bits
        all_n = sharding.gather_ragged(n_det, world)
        all_det = sharding.gather_ragged(det, world)
        all_chk = sharding.gather_ragged(chk, world)
        if rank == 0:
            np.savez(out_path, n_det=all_n.cpu().numpy(), det=all_det.cpu().numpy(), chk=all_chk.cpu().numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_a_single_rank(ra, tmp_path):
    """SURVEY.md 8e on the product: 2 ranks run rp_batch_detect on their stream shards and gather; the gathered block
    equals one rank's run over all the streams (n_det, every detection record, score checksums).  With two GPUs visible
    the ranks own one each and gather over RCCL; on a one-GPU box they share GPU 0 and gather over gloo."""
    import torch
    import torch.multiprocessing as mp
    out = str(tmp_path / "gathered.npz")
    backend = "nccl" if _distinct(2) else "gloo"
    mp.spawn(_rank_worker, args=(2, _free_port(), out, backend, _devices(2)), nprocs=2, join=True)
    z = np.load(out)
    pcm_all = _fixture_streams(8)
    ctx = ra.BatchContext(device=0, host_pointers=True)
    tm = _wakeword(ra, ctx)
    cfg = ra.DetectorConfig()
    cfg.threshold = 0.45
    det, n_det, scores, _ = ctx.batch_detect(pcm_all, tm, cfg, max_det=4, want_scores=True)
    assert np.array_equal(z["n_det"], n_det) and n_det.sum() >= 8
    assert z["det"].astype(np.int32).tobytes() == det.view(np.int32).reshape(det.shape[0], 4, 6).tobytes()
    assert np.array_equal(z["chk"], scores.view(np.int32).astype(np.int64).sum(axis=(1, 2)))


def test_batch_detect_sharded_abi_equals_one_call(ra):
    """rp_batch_detect_sharded: one context + one host thread per shard (three contexts -- on three devices when the node
    has them, else sharing -- ragged shards, one of them empty), results gathered into one host block with global stream ids."""
    pcm_all = _fixture_streams(7)
    ctxs = [ra.BatchContext(d) for d in _devices(3)]
    tms = [_wakeword(ra, c) for c in ctxs]
    cfg = ra.DetectorConfig()
    cfg.threshold = 0.45
    one = ra.BatchContext(0)
    det1, n1 = one.batch_detect(pcm_all, _wakeword(ra, one), cfg, max_det=4)
    for cuts in ((0, 3, 5, 7), (0, 7, 7, 7), (0, 0, 2, 7)):
        parts = [pcm_all[cuts[g]:cuts[g + 1]] for g in range(3)]
        det, n_det = ra.batch_detect_sharded(ctxs, tms, parts, cfg, max_det=4)
        assert np.array_equal(n_det, n1) and det.tobytes() == det1.tobytes()
        assert "host memory" in ra.sharded_gather_info()   # round 4: how the call gathered (rp_sharded_gather_info)
        assert [int(d["stream"]) for s in range(7) for d in det[s][:n_det[s]]] == [s for s in range(7) for _ in range(n_det[s])]
    # errors: a context used for two shards, templates that live on another context
    with pytest.raises(ra.RustpotterError):
        ra.batch_detect_sharded([ctxs[0], ctxs[0]], [tms[0], tms[0]], [pcm_all[:3], pcm_all[3:]], cfg)
    with pytest.raises(ra.RustpotterError):
        ra.batch_detect_sharded([ctxs[0], ctxs[1]], [tms[1], tms[0]], [pcm_all[:3], pcm_all[3:]], cfg)
    # shards that disagree in stream length or sample type are refused before the C call would mis-read them
    with pytest.raises(ValueError):
        ra.batch_detect_sharded(ctxs[:2], tms[:2], [pcm_all[:3], pcm_all[3:, :-480]], cfg)
    with pytest.raises(ValueError):
        ra.batch_detect_sharded(ctxs[:2], tms[:2], [pcm_all[:3], (pcm_all[3:] * 32767).astype(np.int16)], cfg)


def test_batch_detect_sharded_device_pointers(ra):
    """Device-pointer form: every shard's PCM on its own device, the gathered block on the first context's device -- the
    second shard's results cross to it with hipMemcpyPeerAsync (over xGMI when the two contexts sit on different GPUs)."""
    import torch
    pcm_all = _fixture_streams(6)
    devs = _devices(2)
    ctxs = [ra.BatchContext(d, host_pointers=False) for d in devs]
    tms = [_wakeword(ra, c) for c in ctxs]
    cfg = ra.DetectorConfig()
    cfg.threshold = 0.45
    N = pcm_all.shape[1]
    parts = [torch.from_numpy(pcm_all[:2].copy()).to("cuda:%d" % devs[0]), torch.from_numpy(pcm_all[2:].copy()).to("cuda:%d" % devs[1])]
    det = torch.zeros((6, 4, 6), dtype=torch.int32, device="cuda:%d" % devs[0])
    n_det = torch.zeros((6,), dtype=torch.int32, device="cuda:%d" % devs[0])
    for d in set(devs):
        torch.cuda.synchronize(d)
    ra.batch_detect_sharded_dev(ctxs, tms, [p.data_ptr() for p in parts], [2, 4], N, N, cfg, det.data_ptr(), n_det.data_ptr(), 4)
    # round 4: the call says per shard whether its results crossed by a direct peer write (xGMI) or a staged copy
    info = ra.sharded_gather_info()
    assert info.startswith("gather onto device %d:" % devs[0]) and "shard 1 (device %d): " % devs[1] in info
    assert ("same device" in info) == (devs[0] == devs[1]) and (devs[0] == devs[1] or "peer access" in info or "staged" in info)
    with pytest.raises(ValueError):
        ra.batch_detect_sharded_dev(ctxs, tms, [parts[0].data_ptr()], [2, 4], N, N, cfg, det.data_ptr(), n_det.data_ptr(), 4)
    one = ra.BatchContext(0)
    det1, n1 = one.batch_detect(pcm_all, _wakeword(ra, one), cfg, max_det=4)
    assert np.array_equal(n_det.cpu().numpy(), n1)
    assert det.cpu().numpy().tobytes() == det1.view(np.int32).reshape(6, 4, 6).tobytes()


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (the shape of the driver's command): the parent starts the two ranks
    before touching the GPU and relays rank 0's JSON line.  On a one-GPU box the ranks share the device (dry run)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    import torch
    if torch.cuda.device_count() < 2:
        env["RP_BENCH_OVERSUBSCRIBE"] = "1"   # one GPU: the two ranks share it over gloo (a launch-path dry run)
    else:
        env.pop("RP_BENCH_OVERSUBSCRIBE", None)  # two or more: one rank per GPU over RCCL, exactly the driver's run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "2048", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["steps"] == 2 and j["scaling"] == "weak"
    assert j["roofline_other"]["kernel"] != j["roofline"]["kernel"]
    for r_ in (j["roofline"], j["roofline_other"]):   # frac = the largest pipe fraction, a fraction of a roof (round 4)
        assert r_["bound"] in r_["pipes"] and 0.0 < r_["frac"] <= 1.0 and r_["frac"] == max(v for k, v in r_["pipes"].items() if k != "valu_flops_ref")
    # the self-proving record of a multi-rank line: the world size the process group reports, one device record per rank
    c = j["config"]
    assert c["rccl_world_size"] == 2 and [e["rank"] for e in c["rank_devices"]] == [0, 1] and c["gather_ms_per_step"]["mean"] > 0
    assert c["distinct_devices"] == (2 if torch.cuda.device_count() >= 2 else 1) and c["build"] == "gfx950"
    if torch.cuda.device_count() < 2:
        assert j["oversubscribed"]["devices"] == torch.cuda.device_count()
        # without the override a node with too few GPUs is refused, not silently oversubscribed
        env.pop("RP_BENCH_OVERSUBSCRIBE")
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "512"], capture_output=True,
                            text=True, timeout=300, env=env, cwd=ROOT)
        assert r2.returncode != 0 and "RP_BENCH_OVERSUBSCRIBE" in r2.stderr


def test_bench_c4_preset_is_strong_scaling():
    """`bench.py --gpus 2 --config C4`: BASELINE config C4 as stated -- 65 536 streams x 64 templates SPLIT over the ranks by
    shard_bounds, RCCL gather of the per-stream results; the line says strong scaling and the world size it ran with."""
    import torch
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["RP_BENCH_OVERSUBSCRIBE"] = "1"
    else:
        env.pop("RP_BENCH_OVERSUBSCRIBE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C4", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["world_size"] == 2
    assert j["config"]["workload"].startswith("C4: 65536 synthetic") and "split over 2 rank(s)" in j["config"]["workload"]
    assert j["config"]["streams_per_gpu"] == 32768 and j["config"]["templates"] == 64
    # 65 536 streams x 297 windows per step, whatever the number of ranks
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 - 65536 * 297) < 1.0
    assert j["config"]["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")


def test_rccl_smoke_world_size_1():
    """The exchange bench.py does after a pass, on the real backend: init_process_group("nccl") (= RCCL), all_gather of a
    per-stream int32 tensor, barrier, MAX all_reduce -- one rank, so that it runs on a one-GPU box too."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_smoke.py")], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_contexts_on_one_device_use_large_lds_kernels(ra):
    """The >64 KB dynamic-LDS attribute is set per (device, kernel), not once per process: a second context still
    launches the generic DTW kernel with a long template (LDS above 64 KB)."""
    K, L = 24, 600   # (64 + L - 1) * 25 * 4 B = 66 KB of frames alone
    rng = np.random.default_rng(1)
    templates = [rng.standard_normal((L, K)).astype(np.float32)]
    mf = rng.standard_normal((2, L + 70, K)).astype(np.float32)
    outs = []
    for _ in range(2):
        ctx = ra.BatchContext(0)
        scores, _, _ = ctx.dtw_scores(mf, ra.Templates(ctx, templates))
        outs.append(scores)
    assert np.array_equal(outs[0], outs[1]) and np.isfinite(outs[0]).all()
    ref, _ = orc.score_stream(mf[0][:L + 3], templates)
    assert np.allclose(outs[0][0][:ref.shape[0]], ref, rtol=1e-5, atol=0)


def test_very_long_templates(ra):
    """Templates far beyond the usual 1-2 s: 25 s (the register kernels with > 64 KB of LDS), 50 s (too long for them: the
    generic kernel), and 90 s (refused with a text that names the limit, not a launch failure)."""
    ctx = ra.BatchContext(0)
    K = 5
    for L in (2500, 5000):
        rng = np.random.default_rng(L)
        templates = [rng.standard_normal((L, K)).astype(np.float32), rng.standard_normal((L - 7, K)).astype(np.float32)]
        mf = rng.standard_normal((2, L + 20, K)).astype(np.float32)
        sc, _, _ = ctx.dtw_scores(mf, ra.Templates(ctx, templates))
        ref, _ = orc.score_stream(mf[1][:L + 2], templates)
        assert np.allclose(sc[1][:ref.shape[0]], ref, rtol=1e-5, atol=0)
    with pytest.raises(ra.RustpotterError, match="too long for the device kernels"):
        ra.Templates(ctx, [np.ones((9000, K), np.float32)])


# ------------------------------------------------------------------ NULL arguments with live handles
def test_null_arguments_with_live_handles(ra):
    """config / pcm / det == NULL next to a valid context and template set is an error return, not a crash."""
    import ctypes as C
    L = ra.load_library()
    ctx = ra.BatchContext(0)
    tm = _wakeword(ra, ctx)
    cfg = ra.DetectorConfig()._c()
    pcm = np.zeros((1, 4800), np.float32)
    det = np.zeros((1, 4), dtype=[("a", "<i4", 6)])
    n_det = np.zeros(1, np.int32)
    ok = L.rp_batch_detect_fmt(ctx._h, pcm.ctypes.data, 3, 1, 4800, 4800, tm._h, C.byref(cfg), det.ctypes.data, n_det.ctypes.data, 4, None, None)
    assert ok == 0
    assert L.rp_batch_detect_fmt(ctx._h, pcm.ctypes.data, 3, 1, 4800, 4800, tm._h, None, det.ctypes.data, n_det.ctypes.data, 4, None, None) == -1
    assert b"null" in L.rp_last_error()
    assert L.rp_batch_detect_fmt(ctx._h, None, 3, 1, 4800, 4800, tm._h, C.byref(cfg), det.ctypes.data, n_det.ctypes.data, 4, None, None) == -1
    assert L.rp_batch_detect_fmt(ctx._h, pcm.ctypes.data, 3, 1, 4800, 4800, tm._h, C.byref(cfg), None, n_det.ctypes.data, 4, None, None) == -1
    h = C.c_void_p()
    assert L.rp_stream_batch_new(ctx._h, tm._h, None, 4, 1, C.byref(h)) == -1
    assert L.rp_detect_scan(ctx._h, pcm.ctypes.data, None, 1, 10, 5, None, 0, None, 0, det.ctypes.data, n_det.ctypes.data, 4) == -1


# ------------------------------------------------------------------ BASELINE configs at their sizes
def test_c2_full_size_vs_oracle(ra):
    """BASELINE config C2 (1 024 streams x 8 templates, 4 s streams) at size: probabilities, aggregate = row maximum,
    no detection on noise, bit-reproducible, and 16 sampled streams x ALL 297 windows against the oracle at 1e-5."""
    import torch
    S, T, N, L, K = 1024, 8, 64000, 100, 5
    ctx = ra.BatchContext(device=0, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    templates = orc.synth_templates(SEED, T, L, K)
    tmpl = ra.Templates(ctx, templates)
    nf = ra.mfcc_num_frames(N)
    n_win = nf - L + 1
    pcm = torch.empty((S, N), dtype=torch.float32, device="cuda")
    ctx.synth_dev(SEED, 0, S, N, N, pcm.data_ptr())
    scores = torch.empty((S, n_win, T), dtype=torch.float32, device="cuda")
    agg = torch.empty((S, n_win), dtype=torch.float32, device="cuda")
    det = torch.zeros((S, 4, 6), dtype=torch.int32, device="cuda")
    n_det = torch.zeros((S,), dtype=torch.int32, device="cuda")
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, scores.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    assert n_win == 297 and bool(torch.isfinite(scores).all()) and float(scores.min()) > 0.0 and float(scores.max()) < 1.0
    assert torch.equal(agg, scores.max(dim=2).values) and int(n_det.sum()) == 0
    s2 = torch.empty_like(scores)
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, s2.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(scores, s2)
    for s in list(range(0, S, 73)) + [S - 1]:
        ref_pcm = orc.synth_pcm(SEED, s, N)
        ref_s, _ = orc.score_stream(orc.mfcc_stream(ref_pcm, K), templates)
        got = scores[s].cpu().numpy()
        assert ref_s.shape == got.shape and np.all(np.abs(got - ref_s) <= 1e-5 * np.abs(ref_s)), s


def test_c3_sampled_streams_all_windows_vs_oracle(ra):
    """BASELINE config C3 at size (65 536 x 8): 16 sampled streams x all 297 windows against the oracle at 1e-5 (the
    size-independent properties are test_full_size_properties)."""
    import torch
    S, T, N, L, K = 65536, 8, 64000, 100, 5
    ctx = ra.BatchContext(device=0, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    templates = orc.synth_templates(SEED, T, L, K)
    tmpl = ra.Templates(ctx, templates)
    nf = ra.mfcc_num_frames(N)
    n_win = nf - L + 1
    pcm = torch.empty((S, N), dtype=torch.float32, device="cuda")
    ctx.synth_dev(SEED, 0, S, N, N, pcm.data_ptr())
    scores = torch.empty((S, n_win, T), dtype=torch.float32, device="cuda")
    agg = torch.empty((S, n_win), dtype=torch.float32, device="cuda")
    det = torch.zeros((S, 4, 6), dtype=torch.int32, device="cuda")
    n_det = torch.zeros((S,), dtype=torch.int32, device="cuda")
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), 4, scores.data_ptr(), agg.data_ptr())
    torch.cuda.synchronize()
    picks = [0, 1, 63, 64, 4095, 4096, 21845, 21846, 32767, 32768, 43690, 43691, 54321, 65000, 65534, 65535]
    for s in picks:
        ref_s, ref_a = orc.score_stream(orc.mfcc_stream(orc.synth_pcm(SEED, s, N), K), templates)
        got = scores[s].cpu().numpy()
        assert ref_s.shape == got.shape and np.all(np.abs(got - ref_s) <= 1e-5 * np.abs(ref_s)), s
        assert np.all(np.abs(agg[s].cpu().numpy() - ref_a) <= 1e-5 * np.abs(ref_a)), s


def test_c5_full_size_model_forward(ra):
    """BASELINE config C5 at size: B = 65 536 rows x 3 120 features through the Small stack 3120 -> 32 -> 16 -> 2.
    Finite logits, bit-reproducible, a row's logits do not depend on the batch it is in, 256 sampled rows against the f32
    oracle at 1e-5 (f32 callers) and against the bf16-rounding oracle at 1e-3 (bf16 MFMA)."""
    import torch
    B, dims = 65536, [3120, 32, 16, 2]
    rng = np.random.default_rng(5)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    bs = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(3)]
    ctx = ra.BatchContext(device=0, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    model = ra.Model(ctx, ws, bs)
    gen = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((B, dims[0]), dtype=torch.float32, device="cuda", generator=gen)
    rows = np.unique(np.concatenate([[0, 1, 15, 16, 127, 128, B - 1], np.random.default_rng(2).integers(0, B, 256)]))
    xs = x[torch.from_numpy(rows).cuda()].contiguous()
    for prec, tol, bf in (("f32", 1e-5, False), ("bf16", 1e-3, True)):
        out = torch.empty((B, 2), dtype=torch.float32, device="cuda")
        out2 = torch.empty_like(out)
        ctx.mlp_dev(model, x.data_ptr(), B, prec, out.data_ptr())
        ctx.mlp_dev(model, x.data_ptr(), B, prec, out2.data_ptr())
        small = torch.empty((len(rows), 2), dtype=torch.float32, device="cuda")
        ctx.mlp_dev(model, xs.data_ptr(), len(rows), prec, small.data_ptr())
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all()) and torch.equal(out, out2)
        got = out[torch.from_numpy(rows).cuda()].cpu().numpy()
        assert np.array_equal(got, small.cpu().numpy())  # batch invariance
        ref = orc.mlp_forward(xs.cpu().numpy(), ws, bs, bf16_layer1=bf)
        assert np.allclose(got, ref, rtol=tol, atol=tol), (prec, np.abs(got - ref).max())
