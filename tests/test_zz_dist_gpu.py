"""Two ranks (gloo, both on the one GPU of the test box) through the real path: tiny DTTNet forward +
PIT loss + backward on the HIP kernels, GradBucketReducer (bucket all-reduce from grad hooks),
ClipAdamW on the bucket views -- against one process over the concatenated batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2, precision="fp32")


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "speech-separation_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run_steps(rank, world, steps=2, sharded=False):
    _setup_paths()
    import dtt_amd
    from dtt_amd.dist import GradBucketReducer, ShardedClipAdamW
    from helpers import make_batch, to_dev
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    model = dtt_amd.DTTNetModel(**KW).to(dev)
    if sharded:
        # reduce-scatter -> clip + AdamW on this rank's shard -> all-gather (ShardedClipAdamW); replicas start equal
        opt = ShardedClipAdamW(model, lr=1e-3, weight_decay=1e-5, max_norm=0.5)

        class _NoReducer:
            buckets = [0, 0]
            zero_grad = staticmethod(opt.zero_grad)
            finish = staticmethod(lambda: None)
        red = _NoReducer()
    else:
        red = GradBucketReducer(model, bucket_bytes=64 << 10)     # several buckets
        opt = dtt_amd.ClipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5, max_norm=0.5)
    crit = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())
    full = make_batch(4, 800, 64, 16, seed=23)
    per = 4 // world
    sl = slice(rank * per, (rank + 1) * per)

    def cut(v):
        return v[sl] if torch.is_tensor(v) else v
    batch = {k: ([{kk: cut(vv) for kk, vv in d.items()} for d in v] if isinstance(v, list) else cut(v))
             for k, v in full.items()}
    batch = to_dev(batch, dev)
    norms, grads = [], None
    for _ in range(steps):
        b = dict(batch)
        red.zero_grad()
        b.update(model(**b))
        b.update(crit(**b))
        b["loss"].backward()
        red.finish()
        if grads is None and not sharded:
            grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
        norms.append(float(opt.step()))
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}, norms, len(red.buckets), grads or {}


def _worker(rank, world, port, q, sharded=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if os.environ.get("DTT_DIST_TRACE"):                      # debugging aid: where does a stuck rank sit
        import faulthandler
        f = open(os.path.join(ROOT, "gpurun_out", f"dist_rank{rank}.trace"), "w")
        faulthandler.dump_traceback_later(45, exit=True, file=f)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sd, norms, nb, grads = _run_steps(rank, world, sharded=sharded)
        if rank == 0:
            q.put(({k: v.numpy() for k, v in sd.items()}, norms, nb, {k: v.numpy() for k, v in grads.items()}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                         # the parent must not wait for a result that never comes
        import traceback
        q.put(traceback.format_exc())
        raise


def test_two_ranks_match_one_process_over_the_whole_batch():
    world, port = 2, 31000 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=110)
    assert not isinstance(res, str), res
    got, norms, nb, grads = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nb >= 2
    ref, ref_norms, _, ref_grads = _run_steps(0, 1)
    # step 1: same weights, only the batch summation order differs; step 2 starts from weights that already
    # differ by Adam's lr-sized rounding noise (see the parameter bound below)
    for a, b, tol in zip(norms, ref_norms, (1e-4, 3e-3)):
        assert abs(a - b) <= tol * abs(b), (norms, ref_norms)
    # the averaged gradients of step 1: only the summation order over the batch differs
    for k, v in ref_grads.items():
        g = torch.from_numpy(grads[k])
        assert float((g - v).norm()) <= 1e-4 * float(v.norm()) + 1e-7, k
    # parameters after two clipped AdamW steps: Adam's normalised update turns rounding-level gradient
    # differences of near-zero entries into differences of the order of lr, so the bound is lr-sized
    for k, v in ref.items():
        d = (torch.from_numpy(got[k]) - v).abs()
        assert float(d.max()) <= 2.5e-3, (k, float(d.max()))


def _one_step_probe(rank, world, sharded):
    """ONE clipped AdamW step; returns what is LINEAR in the gradient (first moments), the parameter deltas, the
    gradients and -- sharded -- a checksum of the whole flat parameter buffer after the all-gather."""
    _setup_paths()
    import dtt_amd
    from dtt_amd.dist import ShardedClipAdamW
    from helpers import make_batch, to_dev
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    model = dtt_amd.DTTNetModel(**KW).to(dev)
    if sharded:
        opt = ShardedClipAdamW(model, lr=1e-3, weight_decay=1e-5, max_norm=0.5)
    else:
        opt = dtt_amd.ClipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5, max_norm=0.5)
    crit = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())
    full = make_batch(4, 800, 64, 16, seed=23)
    per = 4 // world
    sl = slice(rank * per, (rank + 1) * per)

    def cut(v):
        return v[sl] if torch.is_tensor(v) else v
    batch = to_dev({k: ([{kk: cut(vv) for kk, vv in d.items()} for d in v] if isinstance(v, list) else cut(v))
                    for k, v in full.items()}, dev)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    opt.zero_grad()
    b = dict(batch)
    b.update(model(**b))
    b.update(crit(**b))
    b["loss"].backward()
    norm = float(opt.step())
    torch.cuda.synchronize()
    out = {"norm": norm,
           "delta": {k: (p.detach() - before[k]).cpu().numpy() for k, p in model.named_parameters()}}
    if sharded:
        out["moments"] = {k: (int(off), m.detach().cpu().numpy().copy(), v.detach().cpu().numpy().copy())
                          for k, (off, m, v) in opt.state_shard().items()}
        out["flat_p"] = opt.flat_p.detach().cpu().numpy().copy()
        out["offsets"] = {}
        o = 0
        for k, p in model.named_parameters():
            if p.requires_grad:
                out["offsets"][k] = o
                o += p.numel()
    else:
        out["moments"] = {k: (opt.state[p]["exp_avg"].detach().cpu().numpy().copy(),
                              opt.state[p]["exp_avg_sq"].detach().cpu().numpy().copy())
                          for k, p in model.named_parameters()}
        out["grad"] = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}
    return out


def _probe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        q.put((rank, _one_step_probe(rank, world, True)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


def test_sharded_optimizer_step_two_ranks_match_one_process():
    """reduce-scatter -> sum-of-squares of the shard + all-reduce of one float -> clip + AdamW on the shard ->
    all-gather (ShardedClipAdamW, two gloo ranks on the one GPU) against one process with the replicated ClipAdamW
    over the whole batch, after ONE step, on quantities that CAN fail: the first moments of every rank's shard
    (linear in the clipped gradient: reduce-scatter, segment offsets and the global clip coefficient), the second
    moments, the parameter deltas where |g| >> eps (there Adam's first step is -lr * sign(g): the update itself),
    that every parameter tensor moved, and that both ranks hold bit-identical parameters after the all-gather."""
    import numpy as np
    world, port = 2, 33000 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, res = q.get(timeout=110)
        assert not isinstance(res, str), res
        got[r] = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _one_step_probe(0, 1, False)
    lr = 1e-3
    assert abs(got[0]["norm"] - ref["norm"]) <= 1e-4 * abs(ref["norm"]) and got[0]["norm"] == got[1]["norm"]
    # (1) both ranks hold bit-identical parameters: the all-gather delivered the other rank's shard
    assert np.array_equal(got[0]["flat_p"], got[1]["flat_p"])
    # (2) moments of each rank's segments against the replicated optimizer's, at the segment's offset
    covered = 0
    for r in range(world):
        for name, (off, m, v) in got[r]["moments"].items():
            a = off - got[r]["offsets"][name]                       # position of the segment inside its parameter
            rm, rv = ref["moments"][name]
            rm, rv = rm.reshape(-1)[a:a + m.size], rv.reshape(-1)[a:a + v.size]
            assert m.size > 0 and rm.size == m.size, (name, off, a)
            assert np.linalg.norm(m - rm) <= 1e-4 * np.linalg.norm(rm) + 1e-10, ("exp_avg", name, r)
            assert np.linalg.norm(v - rv) <= 2e-4 * np.linalg.norm(rv) + 1e-14, ("exp_avg_sq", name, r)
            covered += m.size
    assert covered == sum(v.size for v in ref["delta"].values())   # the shards tile every parameter exactly once
    # (3) parameter deltas: the first AdamW step is -lr * gc / (|gc| + eps) with gc = clip coefficient * g (bias
    # corrections cancel; the decay term lr * wd * p is ~1e-8): checked where |gc| >> eps, i.e. where rounding-level
    # differences of g cannot move the result
    coef = min(1.0, 0.5 / (ref["norm"] + 1e-6))
    moved = 0
    for name, d_ref in ref["delta"].items():
        gc = ref["grad"][name] * coef
        big = np.abs(gc) > 1e-6
        d = got[0]["delta"][name]
        if big.any():
            want = -lr * gc[big] / (np.abs(gc[big]) + 1e-8)
            assert np.abs(d_ref[big] - want).max() <= 2e-2 * lr, name           # the replicated step is AdamW's
            assert np.abs(d[big] - want).max() <= 2e-2 * lr, name               # ... and so is the sharded one
            moved += 1
        if np.abs(gc).max() > 0:
            assert np.abs(d).max() > 0, name                                    # nothing with a gradient stood still
    assert moved >= 0.5 * len(ref["delta"]), moved
    # and a single process through the sharded class (world 1: the shard is everything) follows the replicated step
    one = _one_step_probe(0, 1, True)
    assert abs(one["norm"] - ref["norm"]) <= 1e-5 * abs(ref["norm"])
    for name, (off, m, v) in one["moments"].items():
        assert np.linalg.norm(m.reshape(-1) - ref["moments"][name][0].reshape(-1)) <= 1e-5 * np.linalg.norm(ref["moments"][name][0]) + 1e-10, name


def test_reducer_async_path_over_rccl_one_rank():
    """The RCCL ("nccl") leg of GradBucketReducer -- asynchronous bucket all-reduces issued from the grad hooks, waits in
    finish() -- on the one GPU a test box has: a one-rank group with the reducer told the world is 2, so every
    collective runs (ReduceOp.AVG over one rank is the identity: gradients unchanged, one collective per bucket).  Own process:
    the process group must not leak into the other tests."""
    import subprocess
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_single_rank_probe.py")
    r = subprocess.run([sys.executable, "-u", script], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + "\n---- stderr (head) ----\n" + r.stderr[:3000] + "\n---- stderr (tail) ----\n" + r.stderr[-2000:]
    assert "rccl single-rank probe ok" in r.stdout
