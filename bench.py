#!/usr/bin/env python3
"""bench.py -- 10 ms-frame MFCC+DTW scorings/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path (MFCC of every 10 ms frame, T banded DTWs +
score_mode aggregation per window, detection state machine) over S synthetic 16 kHz f32
streams that are already resident in HBM.  One "scoring" = one (stream, 10 ms frame)
step in steady state = one scored window.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over torch.distributed (backend nccl == RCCL on ROCm); streams are sharded by
rank (weak scaling: every GPU gets --streams streams), the only exchange is an RCCL all_gather of the
per-stream detection summary at the end of each step.  The ranks are either started by the caller
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: WORLD_SIZE is set) or, when
WORLD_SIZE is not set, by bench.py itself: the parent starts that same command as a child process BEFORE
anything touches the GPU, relays the child's output (rank 0 prints the one JSON line) and exits with its code.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x5EED000000000001
HBM_PEAK = 8.0e12      # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
VALU_PEAK = 157.3e12   # FLOP/s fp32 vector, same table
MFMA_F16_PEAK = 2.5e15  # FLOP/s dense f16 / bf16 MFMA, same table


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process group.  Nothing in this
    process has initialised the GPU (torch.cuda.device_count() does not); the child is a separate program, not an exec."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if have < n:
        if os.environ.get("RP_BENCH_OVERSUBSCRIBE") != "1":
            sys.stderr.write("bench.py: --gpus %d but this node has %d GPU(s); set RP_BENCH_OVERSUBSCRIBE=1 for a dry run in which "
                             "ranks share devices over gloo (reported as oversubscribed, not a scaling number)\n" % (n, have))
            return 2
        env["RP_BENCH_BACKEND"] = "gloo"
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(r.stdout)
    sys.stdout.flush()
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU (BASELINE config C3)")
    ap.add_argument("--templates", type=int, default=8)
    ap.add_argument("--samples", type=int, default=64000, help="samples per stream (4 s @ 16 kHz)")
    ap.add_argument("--template-len", type=int, default=100)
    ap.add_argument("--mfcc-size", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default=None,
                    help="BASELINE.json presets: C2 = 1 024 streams x 8 templates; C3 = 65 536 x 8 (the default workload); C4 = 65 536 streams x 64 "
                         "templates SPLIT over the --gpus ranks (strong scaling, RCCL gather of the per-stream results); C5 = --mode mlp")
    ap.add_argument("--mode", choices=["dtw", "mlp", "stream", "resample"], default="dtw",
                    help="dtw: the headline MFCC+DTW path (default); mlp: BASELINE config C5, wakeword-model forward; "
                         "stream: the same path fed --chunks-per-call 30 ms chunks per call (rp_stream_batch_process); "
                         "resample: 48 kHz -> 16 kHz front-end alone (rp_resample_batch)")
    ap.add_argument("--chunks-per-call", type=int, default=1)
    ap.add_argument("--pcm-format", choices=["f32", "i16"], default="f32", help="--mode resample: sample format of the 48 kHz input")
    ap.add_argument("--channels", type=int, default=1, help="--mode resample: interleaved channels of the 48 kHz input")
    ap.add_argument("--mlp-precision", choices=["f32", "bf16"], default="bf16")
    ap.add_argument("--template-lens", default="", help="comma-separated template lengths in frames (overrides --templates / "
                    "--template-len), e.g. 108,96,90,93,102 = the shape of the reference's oye_casa_g.rpw")
    ap.add_argument("--score-mode", choices=["average", "max", "median", "p25", "p50", "p75", "p80", "p90", "p95"], default="max")
    ap.add_argument("--avg-gate", action="store_true", help="reference defaults: an averaged template and avg_threshold 0.2 -- windows "
                    "whose avg_score is below it are not compared with the sample templates (wakeword_comp.rs:85-93)")
    ap.add_argument("--avg-threshold", type=float, default=0.2, help="with --avg-gate: DetectorConfig.avg_threshold (reference default 0.2)")
    ap.add_argument("--detect-only", action="store_true", help="do not ask for the per-window score arrays: the call returns detections only and "
                    "(ScoreMode::Max) may abandon DTWs that can no longer reach the threshold (same detections; NOT the headline workload, which "
                    "scores every window against every template)")
    ap.add_argument("--full-scores", action="store_true", help="with --avg-gate: compare every window with every template anyway (RP_CTX_FULL_SCORES)")
    args = ap.parse_args()
    args.total_streams = None
    if args.config == "C2":
        args.streams, args.templates = 1024, 8
    elif args.config == "C3":
        args.streams, args.templates = 65536, 8
    elif args.config == "C4":   # strong scaling: the 65 536 streams are split over the ranks (shard by stream)
        args.total_streams, args.templates = 65536, 64
    elif args.config == "C5":
        args.mode = "mlp"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    # RP_BENCH_BACKEND=gloo is a dry-run switch for boxes with fewer GPUs than ranks (ranks then share
    # devices); the driver's runs use the default, nccl == RCCL on ROCm, one rank per GPU.
    backend = os.environ.get("RP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"

    import rustpotter_amd as ra
    from rustpotter_amd import sharding

    if args.mode == "mlp":
        return bench_mlp(args, ra, torch, dist, dev, world, rank, local_rank)

    if args.mode == "stream":
        return bench_stream(args, ra, torch, dist, dev, world, rank, local_rank)
    if args.mode == "resample":
        return bench_resample(args, ra, torch, dist, dev, world, rank, local_rank)

    first_stream = None
    if args.total_streams is not None:   # strong scaling: this rank's contiguous block of the fixed stream set
        lo, hi = sharding.shard_bounds(args.total_streams, world, rank)
        args.streams, first_stream = hi - lo, lo
    S, N, K = args.streams, args.samples, args.mfcc_size
    lens = [int(x) for x in args.template_lens.split(",") if x] or [args.template_len] * args.templates
    T, L = len(lens), max(lens)
    nf = ra.mfcc_num_frames(N)
    n_win = nf - L + 1
    ctx = ra.BatchContext(device=local_rank, host_pointers=False, full_scores=args.full_scores)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # templates: T synthetic utterances, MFCC by the HIP path, whole-matrix mean
    # normalisation, truncated to their length.  Identical arrays are handed to the CPU baseline.
    templates = make_templates(ra, ctx, torch, dev, lens, K)
    avg_t = None
    if args.avg_gate:
        # synthetic averaged template: the frame-wise mean of the templates cut to the shortest one (the reference's
        # MfccAverager aligns them by DTW first; only the amount of work matters here)
        lm = min(lens)
        avg_t = np.ascontiguousarray(np.mean([t[:lm] for t in templates], axis=0, dtype=np.float32), dtype=np.float32)
    tmpl = ra.Templates(ctx, templates, avg=avg_t)

    # resident inputs / outputs
    pcm = torch.empty((S, N), dtype=torch.float32, device=dev)
    ctx.synth_dev(SEED, sharding.weak_first_stream(S, rank) if first_stream is None else first_stream, S, N, N, pcm.data_ptr())
    want_arrays = not (args.avg_gate or args.detect_only)  # the per-window arrays are defined for every window: asking for them keeps every DTW
    scores = torch.empty((S, n_win, T), dtype=torch.float32, device=dev) if want_arrays else None
    agg = torch.empty((S, n_win), dtype=torch.float32, device=dev) if want_arrays else None
    max_det = 4
    det = torch.zeros((S, max_det, 6), dtype=torch.int32, device=dev)
    n_det = torch.zeros((S,), dtype=torch.int32, device=dev)
    cfg = ra.DetectorConfig()
    cfg.score_mode = {"average": 0, "max": 1, "median": 2, "p25": 3, "p50": 4, "p75": 5, "p80": 6, "p90": 7, "p95": 8}[args.score_mode]
    cfg.avg_threshold = args.avg_threshold if args.avg_gate else 0.0  # gate off: exactly T DTWs per scoring

    def step():
        # one C call: mfcc_kernel -> dtw kernel(s) -> aggregate kernel -> scan_kernel on the launch stream
        ctx.batch_detect_dev(pcm.data_ptr(), S, N, N, tmpl, cfg, det.data_ptr(), n_det.data_ptr(), max_det,
                             scores.data_ptr() if want_arrays else None, agg.data_ptr() if want_arrays else None)
        # final per-stream result gather (RCCL over xGMI); shards of a fixed stream set may differ by one stream
        return sharding.gather_per_stream(n_det, world) if first_stream is None else sharding.gather_ragged(n_det, world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    scorings_per_step = (S * world if args.total_streams is None else args.total_streams) * n_win
    value = scorings_per_step * args.steps / dt

    # ---- rooflines, measured live with HIP events on the launch stream (rp_ctx_timing_*: one event pair per launch)
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(max(2, min(args.steps, 5))):
        step()
    torch.cuda.synchronize()
    k_ms = {name: ctx.timing_read(i) for i, name in enumerate(["mfcc", "dtw", "aggregate", "scan"])}
    ctx.timing_enable(False)
    per_gpu_scorings = S * n_win
    W = 5
    # reference-shaped flops of one template DTW (SURVEY.md S8d F_dtw: 3 dot products + sqrt + divide per cell = 2K+7, norms,
    # normalisation) and the flops the kernel executes (unit-length rows: K FMAs + 2 min3 + 1 add per cell; per row and
    # window one ring column: K subtracts, K FMAs, rsqrt, K multiplies, shared by the templates of a chunk)
    def cells(Lt):
        return sum((min(Lt, r + W - 1) - max(1, r - W) + 1) for r in range(1, Lt))
    f_dtw_ref = sum(cells(Lt) * (2 * K + 7) + 2 * Lt * 2 * K + 2 * Lt * K for Lt in lens)
    by_len = {}
    for Lt in lens:
        by_len[Lt] = by_len.get(Lt, 0) + 1
    n_chunks = sum(-(-c // 8) for c in by_len.values())
    f_dtw_exec = sum((Lt - 1) * 2 * W * (2 * K + 3) for Lt in lens) + sum(-(-c // 8) * ((Lt - 1) * (4 * K + 1) + Lt * K) for Lt, c in by_len.items())
    f_mfcc = 13.2e3
    dom = max(("mfcc", "dtw"), key=lambda n: k_ms[n][0])
    # PMC byte / instruction counts per launch come from the committed rocprofv3 passes of this same command (a profiler
    # cannot run inside the timed process); null for any other workload.
    pmc, pmc_src = {}, None
    if not args.avg_gate and args.score_mode == "max":
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
            w = tj["workload"]
            if (w["streams"], w["samples"], w["templates"], w["template_len"], w["mfcc_size"]) == (S, N, T, L, K) and len(set(lens)) == 1:
                for name, d in tj["kernels"].items():
                    for kn in ("mfcc", "dtw"):
                        if kn + "_" in name and "hbm_bytes_per_launch_corrected" in d and (kn != "dtw" or "dtw_mfma" in name or "dtw" not in pmc):
                            pmc[kn] = d
                pmc_src = "profiles/pmc_traffic_latest.json (committed rocprofv3 --pmc passes of this command, not this run)"
        except Exception:
            pmc = {}
    dtw_s, mfcc_s = k_ms["dtw"][0] * 1e-3, k_ms["mfcc"][0] * 1e-3
    simd_cycles = lambda sec: 1024 * sec * 2.4e9  # 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
    dtw_ref_flops, dtw_exec_flops = per_gpu_scorings * f_dtw_ref, per_gpu_scorings * f_dtw_exec
    dtw_bytes = per_gpu_scorings * (4 * K + 4 * (T + 2))
    # which DTW kernel ran (rp_dtw.hip launch_dtw_k5): chunks of 3..8 same-length templates at mfcc_size 5 / band 5 go to the
    # matrix-core kernel unless RP_DTW_MFMA=0
    mfma_on = os.environ.get("RP_DTW_MFMA", "1")[:1] != "0" and min(lens) >= 12
    mfma_wide = mfma_on and K in (13, 16) and all(c >= 3 for c in by_len.values())  # dtw_mfma_wide_kernel (rp_dtw_mfma_wide.hip)
    mfma_used = mfma_wide or (mfma_on and K == 5 and any(c >= 3 for c in by_len.values()))
    # executed arithmetic of dtw_mfma_kernel: per window and column (L columns) three 32x32x16 MFMAs per 32 windows and chunk
    # (3 x 32768 / 32 flops), vector side per cell one v_min3 (2) + one add, per column and lane ~20 flops of frame work (2 lanes)
    def mfma_chunks(c):  # (chunks with eight template slots, chunks with four, templates left to the register kernels)
        full, rem = divmod(c, 8)
        return full + (1 if rem >= 5 else 0), 1 if 3 <= rem <= 4 else 0, rem if rem <= 2 else 0
    f_mfma_matrix = sum((mfma_chunks(c)[0] * 3 + mfma_chunks(c)[1] * 2) * (Lt + 1) * 32768 / 32.0 for Lt, c in by_len.items())
    f_mfma_vector = sum((c - mfma_chunks(c)[2]) * Lt * 2 * W * 3 + (mfma_chunks(c)[0] + mfma_chunks(c)[1]) * Lt * 40 for Lt, c in by_len.items())
    if mfma_wide:  # every template in chunks of eight slots; 3 tiles x (4 k-steps at mfcc_size 16, 3 at 13) MFMAs per column
        ksteps = 4 if K == 16 else 3
        f_mfma_matrix = sum(-(-c // 8) * 3 * ksteps * (Lt + 1) * 32768 / 32.0 for Lt, c in by_len.items())
        f_mfma_vector = sum(c * Lt * 2 * W * 3 + -(-c // 8) * Lt * 8 * K for Lt, c in by_len.items())
    r_dtw = {"bound": "valu", "kernel": "dtw_mfma_wide_kernel" if mfma_wide else "dtw_mfma_kernel" if mfma_used else "dtw_band_kernel" if K == 5 else "dtw_band_wide_kernel", "achieved": dtw_ref_flops / dtw_s / 1e12 if dtw_s else 0.0, "peak": VALU_PEAK / 1e12,
             "unit": "TFLOP/s", "frac": dtw_ref_flops / dtw_s / VALU_PEAK if dtw_s else 0.0,
             "traffic": pmc.get("dtw", {}).get("hbm_bytes_per_launch_corrected"), "traffic_source": pmc_src if "dtw" in pmc else None,
             "avg_launch_ms": k_ms["dtw"][0], "launches_timed": k_ms["dtw"][1],
             "algorithmic_flops_per_launch": dtw_ref_flops, "executed_flops_per_launch": dtw_exec_flops,
             "executed_flop_frac": dtw_exec_flops / dtw_s / VALU_PEAK if dtw_s else 0.0,
             "hbm": {"achieved": dtw_bytes / dtw_s / 1e9 if dtw_s else 0.0, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": dtw_bytes / dtw_s / HBM_PEAK if dtw_s else 0.0, "algorithmic_bytes_per_launch": dtw_bytes},
             "note": "fp32 vector roofline: `achieved`/`frac` price the REFERENCE-shaped flop count (SURVEY.md 8d: 2K+7 flops per band "
                     "cell) against 157.3 TFLOP/s; the register kernel executes fewer (executed_flops: K FMAs + 2 min3 + 1 add per cell), "
                     "so executed_flop_frac is its honest flop fraction and valu_issue_frac the pipe saturation: VALU instructions x 4 "
                     "cycles (packed-f32 issue slot) / (1024 SIMDs x launch time x 2.4 GHz)"}
    if mfma_used:
        # the cosine costs are formed by v_mfma_f32_32x32x16_f16 (f16 two-way splits, f32 accumulate): the reference-shaped flop rate
        # can exceed the VECTOR peak (frac > 1) because those flops no longer run on the vector pipe.  What bounds the kernel is VALU
        # issue (the recurrence: v_min3 x2 + add x2 per cell pair); valu_busy_frac / mfma_busy_frac below are PMC-measured.
        r_dtw["executed_flops_per_launch"] = per_gpu_scorings * f_mfma_vector
        r_dtw["executed_matrix_flops_per_launch"] = per_gpu_scorings * f_mfma_matrix
        r_dtw["executed_flop_frac"] = per_gpu_scorings * f_mfma_vector / dtw_s / VALU_PEAK if dtw_s else 0.0
        r_dtw["mfma_f16_frac"] = per_gpu_scorings * f_mfma_matrix / dtw_s / MFMA_F16_PEAK if dtw_s else 0.0
        r_dtw["note"] = ("dtw_mfma_kernel: the cosine costs of a band column come out of v_mfma_f32_32x32x16_f16 (f16 two-way splits of both "
                         "operands, f32 accumulate), the vector pipe runs the recurrence.  `achieved`/`frac` still price the REFERENCE-shaped "
                         "flop count (SURVEY.md 8d) against the 157.3 TFLOP/s VECTOR peak, as in earlier rounds -- above 1 means the kernel "
                         "beats what any f32 vector formulation of the reference's arithmetic could reach, not that a roof is exceeded.  "
                         "Bound: VALU issue -- valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x shader cycles) and mfma_busy_frac = "
                         "SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x shader cycles), both from the committed PMC passes; mfma_f16_frac = executed "
                         "matrix flops against the 2.5 PFLOP/s dense f16 peak")
    if "dtw" in pmc and "instructions_per_launch" in pmc["dtw"]:
        r_dtw["valu_insts_per_launch"] = pmc["dtw"]["instructions_per_launch"]["SQ_INSTS_VALU"]
        r_dtw["valu_issue_frac"] = 4.0 * r_dtw["valu_insts_per_launch"] / simd_cycles(dtw_s) if dtw_s else 0.0
        if "effective_clock_ghz" in pmc["dtw"]:  # the chip clocks below 2.4 GHz under this load (GRBM_GUI_ACTIVE / duration, same PMC file)
            r_dtw["effective_clock_ghz"] = pmc["dtw"]["effective_clock_ghz"]
            r_dtw["valu_issue_frac_at_effective_clock"] = r_dtw["valu_issue_frac"] * 2.4 / pmc["dtw"]["effective_clock_ghz"]
    for kf in ("valu_busy_frac", "mfma_busy_frac"):
        if "dtw" in pmc and kf in pmc["dtw"]:
            r_dtw[kf] = pmc["dtw"][kf]
    mfcc_bytes, mfcc_flops = S * nf * (640 + 4 * K), S * nf * f_mfcc
    r_mfcc = {"bound": "hbm", "kernel": "mfcc_kernel", "achieved": mfcc_bytes / mfcc_s / 1e9 if mfcc_s else 0.0, "peak": HBM_PEAK / 1e9,
              "unit": "GB/s", "frac": mfcc_bytes / mfcc_s / HBM_PEAK if mfcc_s else 0.0,
              "traffic": pmc.get("mfcc", {}).get("hbm_bytes_per_launch_corrected"), "traffic_source": pmc_src if "mfcc" in pmc else None,
              "avg_launch_ms": k_ms["mfcc"][0], "launches_timed": k_ms["mfcc"][1], "algorithmic_bytes_per_launch": mfcc_bytes,
              "fp32_frac": mfcc_flops / mfcc_s / VALU_PEAK if mfcc_s else 0.0,
              "note": "660 B per frame (640 B of new PCM + 4K B out) against 8 TB/s; fp32_frac = 13.2 kflop per frame against 157.3 TFLOP/s"}
    if "mfcc" in pmc and "instructions_per_launch" in pmc["mfcc"]:
        r_mfcc["valu_insts_per_launch"] = pmc["mfcc"]["instructions_per_launch"]["SQ_INSTS_VALU"]
        r_mfcc["valu_issue_frac"] = 4.0 * r_mfcc["valu_insts_per_launch"] / simd_cycles(mfcc_s) if mfcc_s else 0.0
        if "effective_clock_ghz" in pmc["mfcc"]:
            r_mfcc["effective_clock_ghz"] = pmc["mfcc"]["effective_clock_ghz"]
            r_mfcc["valu_issue_frac_at_effective_clock"] = r_mfcc["valu_issue_frac"] * 2.4 / pmc["mfcc"]["effective_clock_ghz"]
    roofline = dict(r_dtw if dom == "dtw" else r_mfcc)
    roofline["kernels_ms"] = {k: round(v[0], 4) for k, v in k_ms.items()}
    if k_ms["aggregate"][1] == 0:  # no launch of the aggregate pass: ScoreMode::Max ran inside the DTW kernel
        roofline["aggregate_inside_dtw_kernel"] = True
    work_skipped = args.detect_only or (args.avg_gate and not args.full_scores)
    roofline["path_hbm_frac"] = (value / world) * (640 + 4 * (T + 2)) / HBM_PEAK
    roofline["path_valu_frac_fp32"] = (value / world) * (f_mfcc * nf / n_win + f_dtw_ref) / VALU_PEAK

    tag = {(65536, 8): "C3", (8192, 64): "C4 (per-GPU share)", (1024, 8): "C2"}.get((S, T), "custom") if len(set(lens)) == 1 and lens[0] == 100 else "custom"
    if args.total_streams is not None:
        tag = "C4"
    out = {
        "metric": "10ms-frame MFCC+DTW scorings/sec", "value": value, "unit": "scorings/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.total_streams is None else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %s synthetic 16 kHz f32 streams x %d templates%s (%g s streams, L=%s, K=%d, band 5, "
                               "ScoreMode::%s, %s)" % (tag, ("%d" % S) if args.total_streams is None else ("%d" % args.total_streams), T,
                                                       " per GPU" if args.total_streams is None else " split over %d rank(s) by stream" % world,
                                                       N / 16000.0, lens[0] if len(set(lens)) == 1 else "/".join(map(str, lens)), K,
                                                       args.score_mode.capitalize(),
                                                       ("averaged template + avg_threshold %g%s, " % (args.avg_threshold, " (reference default)" if args.avg_threshold == 0.2 else "") +
                                                        ("every window scored anyway" if args.full_scores else "gated windows skipped"))
                                                       if args.avg_gate else ("avg gate off" + (", detect-only call: DTWs that cannot reach threshold 0.5 "
                                                                                               "any more are abandoned" if args.detect_only else ""))),
                   "streams_per_gpu": S, "templates": T, "samples_per_stream": N, "frames_per_stream": nf,
                   "windows_per_stream": n_win, "template_chunks": n_chunks,
                   "parallelism": "streams sharded x%d, RCCL all_gather of detections" % world,
                   "world_size": world, "backend": (backend if world > 1 else "none (one rank)")},
        # work is skipped by design in detect-only / gated runs: pricing the full flop count against the shorter time would
        # print a fraction above 1, so those lines carry the kernel times only
        "roofline": roofline if not work_skipped else None,
        "roofline_other": (r_mfcc if dom == "dtw" else r_dtw) if not work_skipped else None,
    }
    if work_skipped:
        out["kernels_ms"] = roofline["kernels_ms"]
        out["note"] = "work is skipped by design in this mode (gated windows / abandoned DTWs): no roofline fraction is quoted"
    if backend != "nccl" and world > 1:
        out["oversubscribed"] = {"devices": torch.cuda.device_count(), "backend": backend,
                                 "note": "ranks share GPUs: a launch-path dry run, not a scaling measurement"}
    if args.avg_gate:
        # how many windows pass the gate on this input (one extra pass over the averaged template, not timed)
        mf = torch.empty((S, nf, K), dtype=torch.float32, device=dev)
        ctx.mfcc_dev(pcm.data_ptr(), S, N, N, K, mf.data_ptr())
        sc_ = torch.empty((S, n_win, T), dtype=torch.float32, device=dev)
        av_ = torch.empty((S, n_win), dtype=torch.float32, device=dev)
        ag_ = torch.empty((S, n_win), dtype=torch.float32, device=dev)
        ctx.dtw_dev(mf.data_ptr(), S, nf, tmpl, cfg.score_ref, cfg.band_size, cfg.score_mode, 1, sc_.data_ptr(), av_.data_ptr(), ag_.data_ptr())
        torch.cuda.synchronize()
        out["config"]["avg_threshold"] = cfg.avg_threshold
        out["config"]["gate_pass_fraction"] = float((~(av_ < cfg.avg_threshold)).float().mean().item())
        qs = torch.quantile(av_.flatten()[:: max(1, av_.numel() // 4000000)], torch.tensor([0.001, 0.01, 0.1, 0.5, 0.9, 0.99, 0.999], device=dev))
        out["config"]["avg_score_quantiles"] = {"q": [0.001, 0.01, 0.1, 0.5, 0.9, 0.99, 0.999], "avg_score": [round(float(x), 4) for x in qs]}
        del mf, sc_, av_, ag_

    # ---- CPU baseline: the oracle's restatement of the reference algorithm on this host's cores
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.avg_gate and len(set(lens)) == 1:
        from oracle import rp_oracle as orc
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:  # honour a cgroup CPU quota (the GPU box grants 16 of its 256 hardware threads)
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cores = max(1, min(cores, int(int(quota) / int(period))))
        except Exception:
            pass
        secs, sc, _ = orc.bench(SEED, cores, N, templates, threads=cores)  # calibration: 1 stream per core
        rate = sc / secs
        s_cpu = int(max(cores, min(4096 * cores, args.cpu_seconds * rate / n_win)))
        secs, sc, _ = orc.bench(SEED, s_cpu, N, templates, threads=cores)
        out["cpu_baseline"] = {"value": sc / secs, "unit": "scorings/s", "cores": cores, "kind": "port",
                               "sample": "%d of the same synthetic streams x %d templates, %d scorings in %.1f s; C restatement of "
                                         "the reference algorithm (complex FFT-480 per frame, dense mel, 3-dot cosine per DTW cell), "
                                         "not the Rust crate" % (s_cpu, T, sc, secs)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def make_templates(ra, ctx, torch, dev, lens, K):
    """Template t = MFCC (HIP path) of a synthetic utterance seeded SEED+1+t, whole-matrix mean normalisation, cut to lens[t]."""
    import numpy as np
    T, L = len(lens), max(lens)
    n_t = 480 * -(-(L + 3) // 3)
    tp = torch.empty((T, n_t), dtype=torch.float32, device=dev)
    for t in range(T):
        ctx.synth_dev(SEED + 1 + t, 0, 1, n_t, n_t, tp[t].data_ptr())
    tmf = torch.empty((T, ra.mfcc_num_frames(n_t), K), dtype=torch.float32, device=dev)
    ctx.mfcc_dev(tp.data_ptr(), T, n_t, n_t, K, tmf.data_ptr())
    torch.cuda.synchronize()
    return [np.ascontiguousarray((m - m.mean(axis=0, dtype=np.float32))[:lens[t]], dtype=np.float32) for t, m in enumerate(tmf.cpu().numpy())]


def bench_stream(args, ra, torch, dist, dev, world, rank, local_rank):
    """Live serving shape of the same path: S streams per GPU, every call brings --chunks-per-call new
    30 ms chunks per stream (f32, resident in HBM) and returns that call's detections; extractor history,
    MFCC window and detector state stay on the device between calls."""
    S, T, n = args.streams, args.templates, args.chunks_per_call
    ctx = ra.BatchContext(device=local_rank, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lens = [int(x) for x in args.template_lens.split(",") if x] or [args.template_len] * args.templates
    T = len(lens)
    tmpl = ra.Templates(ctx, make_templates(ra, ctx, torch, dev, lens, args.mfcc_size))
    cfg = ra.DetectorConfig()
    cfg.avg_threshold = 0.0
    sb = ra.StreamBatch(ctx, tmpl, cfg, S, max_chunks_per_call=n)
    n_calls = args.warmup + args.steps + 5
    pcm = torch.empty((S, 480 * n * 4), dtype=torch.float32, device=dev)  # 4 distinct calls' worth, cycled
    ctx.synth_dev(SEED, 0, S, pcm.shape[1], pcm.shape[1], pcm.data_ptr())
    det = torch.zeros((S, 4, 6), dtype=torch.int32, device=dev)
    n_det = torch.zeros((S,), dtype=torch.int32, device=dev)
    calls = [0]

    def step():
        off = (calls[0] % 4) * 480 * n
        calls[0] += 1
        sb.process_dev(pcm.data_ptr() + 4 * off, 3, n, pcm.shape[1], det.data_ptr(), n_det.data_ptr(), 4)

    # fill the window first so that every timed call scores complete windows
    for _ in range(-(-max(lens) // (3 * n)) + 1 + args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    k_ms = {name: round(ctx.timing_read(i)[0], 4) for i, name in enumerate(["mfcc", "dtw", "aggregate", "scan"])}
    ms = dt / args.steps * 1e3
    res = {"metric": "10ms-frame MFCC+DTW scorings/sec (streaming calls)", "value": S * 3 * n * world * args.steps / dt,
           "unit": "scorings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%d live streams x %d templates (%s frames) per GPU, %d chunk(s) of 30 ms per call" % (S, T, "/".join(str(x) for x in sorted(set(lens))), n),
                      "real_time_factor": 30.0 * n / ms, "kernels_ms": k_ms}}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def bench_resample(args, ra, torch, dist, dev, world, rank, local_rank):
    """The sample-rate converter in front of the path: S streams of 4 s at 48 kHz f32 -> 16 kHz."""
    if world > 1:
        raise SystemExit("--mode resample is a single-GPU measurement (launch it with --gpus 1)")
    fs = 48000
    S = min(args.streams, 16384)  # 16384 x 4 s x 48 kHz f32 = 12.6 GB in, 4.2 GB out, + the staged copy
    fi, fo = ra.resampler_frame_lengths(fs)
    n = args.samples * 3
    nch = n // fi
    ctx = ra.BatchContext(device=local_rank, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ch = args.channels
    pcm = torch.empty((S, n), dtype=torch.float32, device=dev)
    ctx.synth_dev(SEED, 0, S, n, n, pcm.data_ptr())
    if args.pcm_format == "i16" or ch > 1:
        mono = (pcm * 32767.0).round().to(torch.int16) if args.pcm_format == "i16" else pcm
        pcm = mono.unsqueeze(2).expand(S, n, ch).contiguous().view(S, n * ch)
        del mono
    fmt = 1 if args.pcm_format == "i16" else 3
    out = torch.empty((S, nch * fo), dtype=torch.float32, device=dev)

    def step():
        ctx.resample_dev(pcm.data_ptr(), fmt, ch, fs, S, n, n * ch, out.data_ptr(), nch * fo)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ms, _n = ctx.timing_read(5)
    alg = S * nch * (fi * ch * (2 if fmt == 1 else 4) + fo * 4)
    gemm = os.environ.get("RP_RESAMPLE_GEMM") == "1"
    if gemm:   # the general kernel: one [2*fi x fo] product per output frame on the f32 matrix cores
        flops = S * nch * 2.0 * (2 * fi) * fo
        roof = {"bound": "mfma", "kernel": "resample_mfma_kernel", "achieved": flops / (ms * 1e-3) / 1e12, "peak": 157.3,
                "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 157.3e12, "traffic": None, "avg_launch_ms": ms}
    else:      # 48 kHz: pruned-FFT kernel, ~110 kflop per frame (8 FFT-240 + twiddle / untangle passes)
        roof = {"bound": "hbm", "kernel": "resample48_fft_kernel", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": alg / (ms * 1e-3) / HBM_PEAK, "traffic": None, "avg_launch_ms": ms,
                "valu_frac_fp32": S * nch * 110e3 / (ms * 1e-3) / VALU_PEAK}
    res = {"metric": "resampled 10ms output frames/sec (48 kHz -> 16 kHz)", "value": S * nch * 3 / dt, "unit": "frames/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%d streams x %d samples at 48 kHz %s, %d channel(s)" % (S, n, args.pcm_format, ch)},
           "roofline": roof}
    if rank == 0:
        print(json.dumps(res))


def bench_mlp(args, ra, torch, dist, dev, world, rank, local_rank):
    """BASELINE config C5: B = 65 536 rows x 3 120 features (F=195 frames x K=16), Small model
    3120 -> 32 -> 16 -> 2 (src/wakewords/nn/wakeword_nn.rs:325-345), rows resident in HBM."""
    import numpy as np
    B, F, K = args.streams, 195, 16
    dims = [F * K, F // 6, F // 12, 2]
    rng = np.random.default_rng(5)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    bs = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(3)]
    ctx = ra.BatchContext(device=local_rank, host_pointers=False)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    model = ra.Model(ctx, ws, bs)
    x = torch.randn((B, dims[0]), dtype=torch.float32, device=dev)
    out = torch.empty((B, dims[-1]), dtype=torch.float32, device=dev)

    def step():
        ctx.mlp_dev(model, x.data_ptr(), B, args.mlp_precision, out.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ms, _n = ctx.timing_read(4)
    alg = B * (dims[0] * 4 + dims[-1] * 4)
    stream = os.environ.get("RP_MLP_STREAM", "1") != "0"
    kname = "mlp_stream_kernel" if stream else "mlp_mfma_kernel"
    # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (a profiler cannot run inside the
    # timed process); null for any other workload
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_c5_latest.json")))
        w = tj["workload"]
        if (w["rows"], w["features"], w["precision"]) == (B, dims[0], args.mlp_precision):
            for name, d in tj["kernels"].items():
                if name.startswith(kname) and "hbm_bytes_per_launch_corrected" in d:
                    traffic = d["hbm_bytes_per_launch_corrected"]
                    traffic_src = "profiles/pmc_c5_latest.json (committed rocprofv3 --pmc passes of this command, not this run)"
    except Exception:
        pass
    res = {"metric": "wakeword-model rows/sec (BASELINE config C5)", "value": B * world * args.steps / dt, "unit": "rows/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16 inputs, f32 accumulate" if args.mlp_precision == "bf16" else "f32", "data": "synthetic",
           "config": {"workload": "C5: %d rows x %d features, MLP %s" % (B, dims[0], "->".join(map(str, dims)))},
           "roofline": {"bound": "hbm", "kernel": kname, "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": alg / (ms * 1e-3) / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                        "avg_launch_ms": ms, "launches_timed": _n, "algorithmic_bytes_per_launch": alg,
                        "note": "12 480 B of features in + 8 B of logits out per row (SURVEY.md 8d) against 8 TB/s"}}
    # ---- CPU baseline: the oracle's restatement of the reference forward (candle's Linear -> ReLU chain, f32) on this host's cores
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import threading
        from oracle import rp_oracle as orc
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                cores = max(1, min(cores, int(int(quota) / int(period))))
        except Exception:
            pass
        xh = x[:4096].cpu().numpy()
        t0 = time.perf_counter()
        orc.mlp_forward(xh[:256], ws, bs)
        per_row = (time.perf_counter() - t0) / 256
        n_cpu = int(max(cores, args.cpu_seconds * cores / per_row))   # rows of the same input, cycled per thread
        per_thread = max(1, n_cpu // cores)

        def work():
            left = per_thread
            while left > 0:
                n = min(left, xh.shape[0])
                orc.mlp_forward(xh[:n], ws, bs)   # ctypes releases the GIL: the threads run on separate cores
                left -= n
        th = [threading.Thread(target=work) for _ in range(cores)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        secs = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": per_thread * cores / secs, "unit": "rows/s", "cores": cores, "kind": "port",
                               "sample": "%d of the same rows through the same model in %.1f s on %d threads; C restatement of the reference "
                                         "forward (f32 Linear -> ReLU chain), not the Rust crate / candle" % (per_thread * cores, secs, cores)}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
