"""Multi-GPU paths (shard by stream, one gather of per-stream results): two ranks sharing one GPU over gloo equal a single rank,
rp_batch_detect_sharded (host and device pointers), bench.py starting its own ranks and its C4 preset (strong scaling), the one-rank RCCL smoke."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import rpw_py
import simstream

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = simstream.GOLDEN


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
