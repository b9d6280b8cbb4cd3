#!/usr/bin/env python3
"""Benchmark of the DTTNet separation training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = STFT of the mixture and both sources (GPU DFT GEMM) -> lip encoder
(AV workloads) -> model forward -> PIT-L1 loss -> backward (-> bucketed RCCL
gradient all-reduce when N > 1) -> gradient clip -> AdamW step, on one batch of
synthetic 2-speaker 16 kHz mixtures (+ 25 fps 88x88 lip ROI streams) that is
already resident in HBM.  Weak scaling: the per-GPU batch is fixed.  The default
workload is the north-star target, BASELINE.json configs[3]: DTTNet AV
cross-attention 60M, B = 16 x 2 s, bf16.  The step is recorded once into a HIP
graph and replayed (dtt_amd.GraphedTrainStep); ``--graph 0`` issues it eagerly.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
sys.path.insert(0, ROOT)

SR = 16000
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # BASELINE.json configs[1]: DTTNet audio-only 15M training step, B=64 x 2 s bf16
    "audio15m": dict(model="audio", g=32, n_idp=4, batch=64, seconds=2.0, fwd_gflop_per_item=96.6),
    # configs[3]: DTTNet AV cross-attention 60M, B=16 x 2 s bf16 + 25 fps 88x88 lip ROI (ResNet-18 in the step)
    "xattn60m": dict(model="xattn", g=32, n_idp=4, batch=16, seconds=2.0, fwd_gflop_per_item=271.3),
    # configs[2]: DTTNet AV FiLM 25M, B=64 x 2 s + 25 fps 88x88 lip ROI (frozen ShuffleNetV2 front end in the step)
    "film25m": dict(model="film", g=32, n_idp=4, batch=64, seconds=2.0, fwd_gflop_per_item=96.6 + 7.2 / 3),
    # configs[4]: DTTNet AV cross-attention 200M (g = 64), B=16 x 6 s long-utterance stress (T = 376 frames, H = 1024 LSTMs)
    "xattn200m": dict(model="xattn", g=64, n_idp=4, batch=16, seconds=6.0, fwd_gflop_per_item=2605.0),
}
NAMES = {"audio": "DTTNet audio-only 15M", "xattn": "DTTNet AV cross-attention (ResNet-18 lip encoder in the step)",
         "film": "DTTNet AV FiLM (frozen lip-reading front end in the step)"}


def synth_batch(B, L, seed, dev):
    """Band-limited, slowly modulated noise sources, peak 0.3, generated on the GPU."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    t = torch.arange(L, device=dev, dtype=torch.float32) / SR
    srcs = []
    f = torch.fft.rfftfreq(L, 1.0 / SR).to(dev)
    band = (1.0 / (1.0 + (80.0 / f.clamp(min=1e-3)) ** 8) / (1.0 + (f / 7000.0) ** 8)).sqrt()
    for _ in range(2):
        w = torch.randn(B, L, generator=g, device=dev)
        s = torch.fft.irfft(torch.fft.rfft(w, dim=-1) * band, n=L, dim=-1)
        rate = 2.0 + 6.0 * torch.rand(B, 1, generator=g, device=dev)
        ph = 6.2831853 * torch.rand(B, 1, generator=g, device=dev)
        s = s * (0.5 - 0.5 * torch.cos(6.2831853 * rate * t.unsqueeze(0) + ph))
        srcs.append((0.3 * s / s.abs().amax(dim=-1, keepdim=True).clamp(min=1e-9)).contiguous())
    return srcs


def cpu_baseline(cfg, n_fft, hop, L, seconds_budget=25.0):
    """The oracle (CPU restatement of the reference, oracle/dtt_oracle.py) timed on this host:
    (lip encoder +) forward + PIT-L1 loss + backward, fp32, on a bounded sample (B=2) of the same workload."""
    import oracle
    from oracle.fill import synth_sources, synth_video
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    import dtt_amd
    common = dict(g=cfg["g"], n_sources=2, n_layers=3, n_idp_layers=cfg["n_idp"], n_fft=n_fft, hop_length=hop, n_heads=2)
    if cfg["model"] == "xattn":
        m = dtt_amd.DTTNetVisionModel(video_channels=512, fusion_method="cross_attention_positional", **common)
    elif cfg["model"] == "film":
        return None                 # the frozen front end has no oracle restatement: no CPU figure for cfg3
    else:
        m = dtt_amd.DTTNetModel(**common)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    B = 2
    s1, s2 = synth_sources(B, L, seed=1)
    mix = s1 + s2
    video = synth_video(B, int(L / SR * 25), seed=2) if cfg["model"] == "xattn" else None

    def step():
        spec, phase = oracle.stft_mag_phase(mix, n_fft, hop)
        sources = []
        for s in (s1, s2):
            sp, ph = oracle.stft_mag_phase(s, n_fft, hop)
            sources.append({"audio": s, "spectrogram": sp, "phase": ph})
        if video is not None:
            vf = oracle.resnet18_video_encoder(sd, "video_encoder.", video, training=True)
            out = oracle.dttnet_vision_forward(sd, spec, phase, L, vf, "cross_attention_positional", n_fft=n_fft, hop=hop)
        else:
            out = oracle.dttnet_forward(sd, spec, phase, L, n_fft=n_fft, hop=hop)
        r = oracle.pit_loss(mix, sources, out["preds"])
        r["loss"].backward()
        for v in sd.values():
            v.grad = None

    step()
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < seconds_budget and n < 10):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    what = "lip encoder + " if video is not None else ""
    return {"value": round(B * L / SR / dt, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 {what}fwd+PIT-L1+bwd, B={B} x {L / SR:.0f} s, {n} iterations after 1 warm-up"}


def build_workload(name, args, dev, rank, world, reducer_cls):
    """Model, optimizer, inputs and the step callable of one workload."""
    import dtt_amd
    cfg = dict(WORKLOADS[name])
    B = args.batch or cfg["batch"]
    n_fft, hop = 1024, 256
    L = int(cfg["seconds"] * SR)
    torch.manual_seed(1)                       # train.yaml:56; same weights on every rank
    common = dict(g=cfg["g"], n_sources=2, n_layers=3, n_idp_layers=cfg["n_idp"], n_fft=n_fft, hop_length=hop,
                  n_heads=2, use_checkpoints=True, precision=args.precision)
    if cfg["model"] == "audio":
        model = dtt_amd.DTTNetModel(**common).to(dev)
    elif cfg["model"] == "xattn":
        model = dtt_amd.DTTNetVisionModel(video_channels=512, fusion_method="cross_attention_positional",
                                          **common).to(dev)
    else:
        model = dtt_amd.DTTNetPlusModel(video_embed_dim=512, **common).to(dev)
    Tv = int(cfg["seconds"] * 25)
    gv = torch.Generator(device=dev)
    gv.manual_seed(99 + rank)
    video = None
    if cfg["model"] in ("xattn", "film"):
        video = (torch.rand(B, 2, Tv, 88, 88, generator=gv, device=dev) - 0.421) / 0.165
    criterion = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())
    reducer = reducer_cls(model)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    decay = [p for n, p in named if "bias" not in n]
    no_decay = [p for n, p in named if "bias" in n]
    use_graph = args.graph != "0"
    # clip_grad_norm_(2.0) + AdamW (train.yaml:20-23,59) in two launches
    opt = dtt_amd.ClipAdamW([{"params": decay, "weight_decay": 1e-5}, {"params": no_decay, "weight_decay": 0.0}],
                            lr=1e-4, max_norm=2.0, capturable=use_graph and world == 1)
    pool = []
    for i in range(2):
        s1, s2 = synth_batch(B, L, 1234 + 1000 * rank + i, dev)
        pool.append({"s1": s1, "s2": s2})

    def batch_fn(inp):
        s1, s2 = inp["s1"], inp["s2"]
        mix = s1 + s2
        spec, phase = dtt_amd.stft_mag_phase(mix, n_fft, hop)
        sources = []
        for s in (s1, s2):
            sp, ph = dtt_amd.stft_mag_phase(s, n_fft, hop)
            sources.append({"audio": s, "spectrogram": sp, "phase": ph})
        batch = {"audio": mix, "spectrogram": spec, "phase": phase, "sources": sources, "audio_length": L}
        if video is not None:
            batch["video"] = video
        return batch

    def eager_step(i):
        batch = batch_fn(pool[i % len(pool)])
        reducer.zero_grad()
        batch.update(model(**batch))
        batch.update(criterion(**batch))
        batch["loss"].backward()
        reducer.finish()
        opt.step()
        return batch["loss"]

    mode, step = "eager", eager_step
    if use_graph:
        try:
            gs = dtt_amd.GraphedTrainStep(model, criterion, opt, pool[0], batch_fn, reducer=reducer, warmup=2)
            mode = "hip-graph" if world == 1 else "hip-graph(fwd+bwd) + eager all-reduce/optimizer"

            def step(i, gs=gs):
                return gs(pool[i % len(pool)])
        except Exception as e:                                        # noqa: BLE001
            if args.graph == "1":
                raise
            print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); falling back to eager launches",
                  file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            opt = dtt_amd.ClipAdamW([{"params": decay, "weight_decay": 1e-5},
                                     {"params": no_decay, "weight_decay": 0.0}], lr=1e-4, max_norm=2.0)
    return dict(cfg=cfg, B=B, L=L, n_fft=n_fft, hop=hop, step=step, opt=opt, mode=mode, model=model)


def time_workload(w, steps, warmup, world, dev, backend):
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    w["first_loss"] = None
    for i in range(warmup):
        loss = w["step"](i)
        if i == 0:
            w["first_loss"] = float(loss.detach())      # beside final_loss: the timed steps must train (DESIGN finding 34)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = w["step"](i)
    enqueue = time.perf_counter() - t0          # host time to issue the steps (the GPU runs behind it)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    # host-only issue cost: one step at a time on an IDLE stream (the back-to-back figure above converges to the GPU
    # time once the launch calls block on the full hardware queue, so it says nothing about the host)
    iso = []
    for i in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss = w["step"](steps + i)
        iso.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    w["host_issue_ms"] = sorted(iso)[1] * 1e3
    return elapsed, enqueue, float(loss.detach())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="xattn60m", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", default="auto", choices=["auto", "0", "1"],
                    help="record the step into a HIP graph and replay it (auto: fall back to eager if capture fails)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary audio15m (configs[1]) measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the ranks as a child torchrun (nothing here has touched the GPU yet)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 1000),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # DTT_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks (ranks then share
    # devices); the measured configuration is RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("DTT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from dtt_amd.dist import GradBucketReducer

    w = build_workload(args.workload, args, dev, rank, world, GradBucketReducer)
    cfg, B, launch_mode = w["cfg"], w["B"], w["mode"]
    elapsed, enqueue, loss_val = time_workload(w, args.steps, args.warmup, world, dev, backend)
    host_issue_ms = w["host_issue_ms"]
    w_first = w["first_loss"]

    # SURVEY 8(d) wants the optimizer's share stated beside the step: events on the launch stream around a few
    # extra clip+AdamW calls on the last step's gradients (outside the timed region)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        w["opt"].step()
    e1.record()
    torch.cuda.synchronize()
    opt_ms = e0.elapsed_time(e1) / 5

    roofline = cpu = also = None
    if rank == 0 and not args.no_roofline:
        roofline = dominant_kernel_roofline(cfg, B, args.precision, dev)
    if rank == 0 and world == 1 and not args.no_also and args.workload != "audio15m" and args.batch is None:
        # BASELINE.json configs[1] (the round-1 headline), measured the same way in the same process
        del w
        torch.cuda.empty_cache()
        w2 = build_workload("audio15m", args, dev, rank, world, GradBucketReducer)
        el2, enq2, loss2 = time_workload(w2, max(5, args.steps // 2), 3, world, dev, backend)
        n2 = max(5, args.steps // 2)
        also = {"workload": "audio15m: DTTNet audio-only 15M training step, B=64 x 2 s (BASELINE.json configs[1])",
                "ms_per_step": round(el2 / n2 * 1e3, 3), "value": round(64 * 2.0 / (el2 / n2), 2), "unit": "audio-s/s",
                "launch": w2["mode"], "host_issue_ms_per_step": round(w2["host_issue_ms"], 2),
                "final_loss": round(loss2, 4)}
        del w2
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, 1024, 256, int(cfg["seconds"] * SR))

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * cfg["seconds"] / (elapsed / args.steps)
        step_tflops = 3 * cfg["fwd_gflop_per_item"] * B * world / (elapsed / args.steps) / 1e3
        line = {
            "metric": "separated audio-sec/sec (fwd+bwd, 16 kHz)", "value": round(value, 2), "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {NAMES[cfg['model']]} training step (STFT+fwd+PIT-L1+bwd"
                                   f"+clip+AdamW), B={B}/GPU x {cfg['seconds']:.0f} s, n_fft 1024 hop 256",
                       "global_batch": B * world, "parallelism": f"dp{world}", "launch": launch_mode,
                       "first_loss": None if w_first is None else round(w_first, 4), "final_loss": round(loss_val, 4),
                       "host_issue_ms_per_step": round(host_issue_ms, 2),
                       "host_enqueue_ms_per_step_back_to_back": round(enqueue / args.steps * 1e3, 2),
                       "clip_adamw_ms_per_step": round(opt_ms, 3),
                       "model_tflops_per_s": round(step_tflops, 2),
                       "mfma_frac_whole_step": round(step_tflops / (PEAK_BF16_TFLOPS * world), 4),
                       "also": also},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                         # ranks leave together (rank 0 was busy with the roofline timing)
        dist.destroy_process_group()


def _events_time(launch, n=20, warm=3):
    for _ in range(warm):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                      # torch's current stream == the stream the C ABI launches on (stream_ptr())
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / 1e3


def _time_conv(ops, B, Fq, T, Cc, dt, dev, n=20):
    x = torch.randn(B, Fq, T, Cc, device=dev).to(dt)
    w = (torch.randn(Cc, Cc, 3, 3, device=dev) / (3.0 * Cc ** 0.5))
    bias = torch.zeros(Cc, device=dev)
    wp = ops._pack_fwd(w, "s1", dt)
    import ctypes as C
    from dtt_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    y = torch.empty_like(x)
    part = torch.empty(B * (4 * ((Fq + 15) // 16) * ((T + 15) // 16) + 4) * Cc * 3, dtype=torch.float32, device=dev)

    def launch():                    # stats = NULL: the kernel with its fused statistics, without the merge launch
        check(lib().dtt_conv3x3_s1(ptr(x), ptr(wp), ptr(bias), ptr(y), None, ptr(part), None, B, Fq, T, Cc, Cc,
                                   C.c_float(1e-5), dtype_code(dt), stream_ptr()), "dtt_conv3x3_s1")
    return _events_time(launch, n)


def _time_lstm(ops, B, g, seconds, dt, dev):
    """The time-axis recurrent forward pass of one TC_FC_RNN at the workload's latent size, through the C ABI:
    N = B * heads * 65 sequences of L = T/8 steps, C' = 8g (x-attn latent: 16g channels over 2 heads), H = 2C'.
    The register-stationary kernel (lstm_rs.hip) at H = 512, the LDS-stationary cluster kernel (lstm_cluster.hip)
    at the other widths it covers, else the streaming kernel."""
    import ctypes as C
    from dtt_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    heads, Fq = 2, 65
    Cp, L = 8 * g, (1 + int(seconds * SR) // 256 + 7) // 8
    H, N = 2 * Cp, B * heads * Fq
    R = N * L
    d = dtype_code(dt)
    xg = (torch.randn(R, 8 * H, device=dev) * 0.5).to(dt)
    whh = torch.randn(2, 4 * H, H, device=dev) / H ** 0.5
    y = torch.empty(R, 2 * H, device=dev, dtype=dt)
    rs = ops._rs_ok(N, L, H, d, heads, "auto")
    geom = None if rs else ops._cluster_geometry(N, H, d)
    if rs:
        NT = (N + 31) // 32
        wt = whh.to(dt).contiguous()
        gates = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt)
        cst = torch.empty(2, NT, L, H * 32, device=dev, dtype=torch.float32)
        sync = torch.empty(lib().dtt_lstm_rs_sync_words(N, H), device=dev, dtype=torch.int32)
        name = ("lstm_rs_fwd_deep_kernel (dtt_lstm_rs_fwd: BiLSTM recurrence, W_hh resident in the AGPRs of 8-CU "
                "clusters, h tile once per CU through LDS")

        def launch():
            check(lib().dtt_lstm_rs_fwd(ptr(xg), ptr(wt), ptr(y), ptr(gates), ptr(cst), ptr(sync), N, L, heads, H, d,
                                        stream_ptr()), "dtt_lstm_rs_fwd")
    elif geom is not None:
        NT = (N + 31) // 32
        wt = whh.to(dt).contiguous()
        gates = torch.empty(2, NT, L, 4 * H * 32, device=dev, dtype=dt)
        cst = torch.empty(2, NT, L, H * 32, device=dev, dtype=torch.float32)
        sync = torch.empty(lib().dtt_lstm_cluster_sync_words(geom[2]), device=dev, dtype=torch.int32)
        name = (f"lstm_cluster_fwd_kernel (dtt_lstm_cluster_fwd: BiLSTM recurrence, W_hh resident in LDS over "
                f"{geom[1]} CUs per group, {2 * geom[2]} groups")

        def launch():
            check(lib().dtt_lstm_cluster_fwd(ptr(xg), ptr(wt), ptr(y), ptr(gates), ptr(cst), ptr(sync), N, L, heads, H, d,
                                             stream_ptr()), "dtt_lstm_cluster_fwd")
    else:
        wp = torch.stack([ops._pack_whh_fwd(whh[0], dt), ops._pack_whh_fwd(whh[1], dt)]).contiguous()
        gates = torch.empty(R, 2, 4 * H, device=dev, dtype=dt)
        cst = torch.empty(R, 2, H, device=dev, dtype=torch.float32)
        name = "lstm_fwd_kernel (dtt_lstm_fwd: BiLSTM recurrence, W_hh streamed from L2 every step"

        def launch():
            check(lib().dtt_lstm_fwd(ptr(xg), ptr(wp), ptr(y), ptr(gates), ptr(cst), N, L, heads, H, d, stream_ptr()),
                  "dtt_lstm_fwd")
    sec = _events_time(launch, n=10)
    flops = 2.0 * N * L * 2 * H * 4 * H                    # h_{t-1} W_hh^T, both directions
    return {"kernel": f"{name}; N={N} sequences x L={L} steps, H={H}; 4 launches of this shape + 4 of the "
                      f"frequency-axis shape forward, as many backward, per step)",
            "achieved": round(flops / sec / 1e12, 2), "avg_launch_us": round(sec * 1e6, 1), "flops_per_launch": flops,
            "traffic_key": f"lstm_fwd_H{H}_N{N}_L{L}"}


def _tracked_traffic(key):
    """HBM bytes per launch from the tracked rocprofv3 --pmc passes (profiles/r02_pmc_traffic.json, written by
    tools/pmc_kernels.py from separate FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled for gfx950).  None when
    no such measurement is tracked for this kernel and shape."""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")) as f:
            return json.load(f).get(key, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def dominant_kernel_roofline(cfg, B, precision, dev):
    """Roofline of the kernel with the largest share of the step (rocprofv3, profiles/), timed here alone through
    the C ABI with events on the launch stream.

    AV cross-attention (cfg4 / cfg5): the recurrent kernel of the dual-path LSTMs (H = 2C' = 512 / 1024), MFMA
    bound: 2 * N * L * 2 directions * H * 4H flop per launch for the recurrence (+ the input projection where the
    kernel fuses it).  Audio-only / FiLM (cfg2 / cfg3): the 3x3 conv of U-Net levels 1-3, conv3x3_s1_kernel."""
    from dtt_amd import ops
    dt = torch.bfloat16 if precision == "bf16" else torch.float32
    es = 2 if precision == "bf16" else 4
    out = None
    if cfg["model"] == "xattn":
        out = _time_lstm(ops, B, cfg["g"], cfg["seconds"], dt, dev)
        out["peak"], out["unit"], out["bound"] = PEAK_BF16_TFLOPS, "TFLOP/s", "mfma"
        out["frac"] = round(out["achieved"] / PEAK_BF16_TFLOPS, 4)
        out["traffic"] = _tracked_traffic(out.pop("traffic_key")) if B == cfg["batch"] and precision == "bf16" else None
    Fq, T, Cc = 257, 63, 64
    sec = _time_conv(ops, B, Fq, T, Cc, dt, dev)
    flops = 2.0 * B * Fq * T * Cc * 9 * Cc
    achieved = flops / sec / 1e12
    conv = {"kernel": f"conv3x3_s1_kernel<bf16,{1 if 17 * 4 * B <= 1152 else 2}> (conv3x3 64->64 @ [B,257,63] + fused InstanceNorm statistics)",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "traffic": _tracked_traffic(f"conv3x3_s1_B{B}") if precision == "bf16" else None,
            "avg_launch_us": round(sec * 1e6, 1), "flops_per_launch": flops}
    sec0 = _time_conv(ops, B, 513, 126, 32, dt, dev)
    bytes0 = 2.0 * B * 513 * 126 * 32 * es                # read x + write y once
    c32 = {"kernel": "conv3x3_c32_kernel (conv3x3 32->32 @ [B,513,126])", "bound": "hbm",
           "achieved": round(bytes0 / sec0 / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "frac": round(bytes0 / sec0 / (PEAK_HBM_GBS * 1e9), 4), "avg_launch_us": round(sec0 * 1e6, 1),
           "traffic": _tracked_traffic(f"conv3x3_c32_B{B}") if precision == "bf16" else None,
           "bytes_per_launch": bytes0}
    if out is None:
        conv["also"] = [c32]
        return conv
    out["also"] = [conv, c32]
    return out


if __name__ == "__main__":
    main()
