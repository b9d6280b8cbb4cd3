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
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    decay = [p for n, p in named if "bias" not in n]
    no_decay = [p for n, p in named if "bias" in n]
    use_graph = args.graph != "0"
    sharded = bool(getattr(args, "sharded_optimizer", False)) and world > 1
    if sharded:
        # reduce-scatter -> clip + AdamW on this rank's 1/W of the parameters -> all-gather; the
        # gradients are exchanged by the optimizer, not by a bucket reducer
        from dtt_amd.dist import ShardedClipAdamW, LocalGrads
        reducer = LocalGrads(model)
        opt = ShardedClipAdamW(model, lr=1e-4, weight_decay=1e-5, weight_decay_of=lambda n: 0.0 if "bias" in n else 1e-5,
                               max_norm=2.0)
    else:
        reducer = reducer_cls(model)
        # clip_grad_norm_(2.0) + AdamW (train.yaml:20-23,59) in two launches
        opt = dtt_amd.ClipAdamW([{"params": decay, "weight_decay": 1e-5}, {"params": no_decay, "weight_decay": 0.0}],
                                lr=1e-4, max_norm=2.0, capturable=use_graph)
    pool = []
    for i in range(2):
        s1, s2 = synth_batch(B, L, 1234 + 1000 * rank + i, dev)
        pool.append({"s1": s1, "s2": s2})

    def batch_fn(inp):
        from dtt_amd import ops
        s1, s2 = inp["s1"], inp["s2"]
        mix = s1 + s2
        # the sources' spectrograms feed only the loss: on the side stream (eager issue; a no-op while a graph is
        # being captured), beside the mixture's STFT and the encoder
        sources = []
        if os.environ.get("DTT_BENCH_SIDE_STFT", "1") == "1":
            with ops.video_stream(mix.device) as side:
                for s in (s1, s2):
                    sp, ph = dtt_amd.stft_mag_phase(s, n_fft, hop)
                    sources.append({"audio": s, "spectrogram": sp, "phase": ph})
            spec, phase = dtt_amd.stft_mag_phase(mix, n_fft, hop)
            side.defer(*[t for d in sources for t in (d["spectrogram"], d["phase"])])     # joined by the loss wrapper
        else:
            spec, phase = dtt_amd.stft_mag_phase(mix, n_fft, hop)
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

    mode, step, gs = "eager", eager_step, None
    if not use_graph:
        _eager_pack_plan()
    if use_graph:
        try:
            gs = dtt_amd.GraphedTrainStep(model, criterion, opt, pool[0], batch_fn, reducer=reducer, warmup=2)
            mode = ("hip-graph(fwd+bwd) + sharded optimizer step (reduce-scatter / clip+AdamW on the shard / all-gather)"
                    if sharded else "hip-graph" if world == 1 else
                    "hip-graph (fwd + bwd + bucketed RCCL all-reduce overlapped with backward + optimizer, one launch)"
                    if gs.overlap else "hip-graph(fwd+bwd) + eager all-reduce/optimizer")

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
    return dict(cfg=cfg, B=B, L=L, n_fft=n_fft, hop=hop, step=step, opt=opt, mode=mode, model=model,
                eager_step=eager_step, reducer=reducer, graph_step=gs if step is not eager_step else None)


def _eager_pack_plan():
    """Eager issue: all weight-derived tensors of a step (casts, tap-major conv packs, stacked LSTM weights: ~280
    launches strung through the step) from ONE launch at the start of the forward (ops.weight_plan; same-box A/B
    34.97 / 35.00 -> 34.62 / 34.56 ms).  DTT_PACK_PLAN=0 keeps the per-use packs."""
    from dtt_amd import ops
    if os.environ.get("DTT_PACK_PLAN", "1") != "0":
        ops.weight_plan(True)


def choose_launch(w, world, dev, backend, n=4):
    """--graph auto: the faster of (a) replaying the recorded step and (b) issuing it eagerly, on THIS box.  Eager issue
    runs the weight gradients and the lip encoder on side streams beside the data-gradient chain (ops.wgrad_lane /
    ops.video_lane: -5 ms of 41 at cfg4), which a HIP-graph replay did not reproduce (same time with and without the
    forked branches); the replay wins where the host cannot issue ~1500 launches per step fast enough.  Every rank times
    both (max over ranks) so that all take the same decision."""
    import torch.distributed as dist
    if w.get("graph_step") is None:
        return

    def timed(fn):
        if world > 1:
            dist.barrier()                     # ranks start together: a collective waits for its slowest peer
        for i in range(2):
            fn(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t)
        return el / n * 1e3
    from dtt_amd import ops
    tg = timed(w["step"])
    _eager_pack_plan()                         # the eager step as it would run: its packs from one launch too
    te = timed(w["eager_step"])
    w["calibration"] = {"graph_ms": round(tg, 2), "eager_ms": round(te, 2)}
    if te < 0.98 * tg:
        w["step"] = w["eager_step"]
        w["mode"] = (f"eager issue, weight gradients and lip encoder on side streams (calibration on this box: eager "
                     f"{te:.1f} ms, graph replay {tg:.1f} ms per step)")
    else:
        ops.weight_plan(False)                 # (the recorded step keeps its own plan's buffers)
        w["graph_step"].rearm()
        w["mode"] += f" (calibration on this box: graph replay {tg:.1f} ms, eager {te:.1f} ms per step)"


def time_workload(w, steps, warmup, world, dev, backend):
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    w["first_loss"] = None
    barrier()
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
                    help="1: record the step into a HIP graph and replay it; 0: issue it eagerly; auto: time both on this box "
                         "and take the faster (eager if the capture fails)")
    ap.add_argument("--sharded-optimizer", action="store_true",
                    help="N > 1: reduce-scatter -> clip + AdamW on 1/N of the parameters -> all-gather (ShardedClipAdamW) "
                         "instead of bucketed all-reduce + replicated update")
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
    if args.graph == "auto":
        choose_launch(w, world, dev, backend)
    cfg, B, launch_mode = w["cfg"], w["B"], w["mode"]
    elapsed, enqueue, loss_val = time_workload(w, args.steps, args.warmup, world, dev, backend)
    host_issue_ms = w["host_issue_ms"]
    w_first = w["first_loss"]
    w_cal = w.get("calibration")

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
    if not args.no_roofline:
        # every rank issues the eager steps (with N > 1 they hold the gradient all-reduce); rank 0 reports
        roofline = step_roofline(w, cfg, B, args.precision, args.workload)
    if rank == 0 and world == 1 and not args.no_also and args.workload != "audio15m" and args.batch is None:
        # BASELINE.json configs[1] (the round-1 headline), measured the same way in the same process
        del w
        torch.cuda.empty_cache()
        w2 = build_workload("audio15m", args, dev, rank, world, GradBucketReducer)
        if args.graph == "auto":
            choose_launch(w2, world, dev, backend)
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
                       "launch_calibration_ms_per_step": w_cal,
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


def _tracked_traffic(key):
    """HBM bytes per launch from the tracked rocprofv3 --pmc passes (profiles/r0*_pmc_traffic.json, written by
    tools/pmc_kernels.py from separate FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled for gfx950).  None when
    no such measurement is tracked for this entry point and shape."""
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                v = json.load(f).get(key, {}).get("hbm_bytes_per_launch")
            if v is not None:
                return v
        except (OSError, ValueError):
            pass
    return None


def _traffic_key(row):
    sh = row["shape"]
    if row["entry"] == "dtt_lstm_rs_fwd":
        return f"lstm_fwd_H{sh['H']}_N{sh['N']}_L{sh['L']}"
    if row["entry"] == "dtt_conv3x3_s1" and sh.get("Ci") == 64 and sh.get("F") == 257:
        return f"conv3x3_s1_B{sh['B']}"
    if row["entry"] == "dtt_conv3x3_s1" and sh.get("Ci") == 32:
        return f"conv3x3_c32_B{sh['B']}"
    return "%s_%s" % (row["entry"], "_".join(f"{k}{v}" for k, v in sorted(sh.items()) if k not in ("form", "dtype")))


def step_roofline(w, cfg, B, precision, workload):
    """Roofline that FOLLOWS THE TIME: three eagerly issued steps of the workload just timed, every libdtt_hip call
    bracketed by HIP events on the launch stream (dtt_amd.prof.CallTimer), grouped by (entry point, shape).  The
    headline entry is the one with the largest share of the step; `top` lists the five largest with their shapes,
    launches per step, average duration, algorithmic flops or bytes per launch and fraction of the
    roofline that bounds them; `by_family` sums the step by kernel family.  The same table, all rows, is written by
    tools/step_table.py and tracked under profiles/."""
    from dtt_amd import ops, prof
    step = w["eager_step"]
    # one stream for the table: beside another kernel an entry's bracket measures the pair, not the entry
    lanes = (ops._WGRAD_LANE[0], ops._VIDEO_LANE[0])
    ops.wgrad_lane(False)
    ops.video_lane(False)
    plan_was = ops._PLAN.enabled
    if not plan_was:
        _eager_pack_plan()                     # (a recorded step was timed: the eager probe packs from one launch as well)
    try:
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        n = 3
        with prof.CallTimer() as ct:
            for i in range(n):
                step(i)
        tab = ct.table(n)
    finally:
        ops.wgrad_lane(lanes[0])
        ops.video_lane(lanes[1])
        if not plan_was:
            ops.weight_plan(False)
    rows = tab["rows"]
    for r in rows:
        r["share_of_step"] = round(r["ms_per_step"] / tab["span_ms_per_step"], 4)
    priced = [r for r in rows if "frac" in r]
    top = priced[:5]
    fam = {}
    for r in rows:
        e = r["entry"]
        k = ("lstm recurrence" if "lstm" in e else "3x3 conv (tile kernels)" if e == "dtt_conv3x3_s1" else
             "3x3 conv weight gradient (tile kernels)" if "wgrad" in e else "gemm (Linear / implicit-GEMM conv / DFT)" if e == "dtt_gemm"
             else "InstanceNorm + GELU" if "in_gelu" in e or e == "dtt_in_stats" else "BatchNorm / pool / depthwise (lip encoder)" if "bn_" in e or "pool" in e or "dwconv" in e or "pw_fold" in e
             else "other entry points")
        fam[k] = round(fam.get(k, 0.0) + r["ms_per_step"], 3)
    fam["outside entry points (torch glue, launch gaps of the eager issue)"] = tab["outside_entry_points_ms_per_step"]
    head = dict(top[0]) if top else {}
    if head:
        sh = " ".join(f"{k}={v}" for k, v in head["shape"].items())
        head["kernel"] = f"{head['entry']} [{sh}]: {head['launches_per_step']:g} launches per step, {head['share_of_step'] * 100:.1f} % of the eagerly issued step"
        head["avg_launch_us"] = head["avg_us"]
        head["traffic"] = _tracked_traffic(_traffic_key(head)) if precision == "bf16" and B == cfg["batch"] else None
    for r in top:
        r["traffic"] = _tracked_traffic(_traffic_key(r)) if precision == "bf16" and B == cfg["batch"] else None
    head["measured"] = (f"{n} eager steps of {workload} on ONE stream (side streams off) with HIP events around every C-ABI "
                        f"call on the launch stream (dtt_amd/prof.py); eager step {tab['span_ms_per_step']:.2f} ms, of which {tab['entry_points_ms_per_step']:.2f} ms "
                        f"inside entry points")
    head["top"] = top
    head["by_family_ms_per_step"] = fam
    return head


if __name__ == "__main__":
    main()
