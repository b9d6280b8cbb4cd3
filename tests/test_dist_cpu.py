"""CPU, world_size 2, gloo: the data-parallel gradient bucket reducer
(dtt_amd/dist.py) averages gradients across ranks exactly like a single-process
run over the concatenated batch, with bucket all-reduces issued from grad hooks."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSTEPS = 4


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dtt_amd.dist import GradBucketReducer
    torch.manual_seed(rank)                  # replicas start different: the reducer copies rank 0's state
    net = torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40), torch.nn.Tanh(),
                              torch.nn.Linear(40, 3), torch.nn.Linear(3, 3))   # net[5] gets no gradient in step 2
    net.register_buffer("steps_seen", torch.tensor([7 + rank]))
    red = GradBucketReducer(net, bucket_bytes=1000)        # several buckets
    assert len(red.buckets) >= 2
    assert int(net.steps_seen) == 7
    g = torch.Generator().manual_seed(100)
    x_all = torch.randn(NSTEPS, 2 * world, 12, generator=g)
    out = []
    opt = torch.optim.SGD(net.parameters(), lr=0.0)        # the reference's loop calls optimizer.zero_grad()
    for step in range(NSTEPS):
        # every way a training loop may clear gradients must leave the reducer armed (it re-arms in finish())
        if step == 0:
            red.zero_grad()
        elif step == 1:
            opt.zero_grad()                                # trainer.py:42
        elif step == 2:
            net.zero_grad(set_to_none=False)               # gradients stay views of the buckets, zeroed in place
        else:
            net.zero_grad()
        x = x_all[step, rank * 2:(rank + 1) * 2]
        y = net[:5](x) if step == 1 else net(x)
        y.pow(2).mean().backward()
        red.finish()
        out.append(torch.cat([torch.zeros_like(p).reshape(-1) if p.grad is None else p.grad.reshape(-1)
                              for p in net.parameters()]).clone())
    # a second backward without finish() must be refused, not silently exchanged wrong
    net.zero_grad()
    net(x_all[0, rank * 2:(rank + 1) * 2]).pow(2).mean().backward()
    try:
        net(x_all[0, rank * 2:(rank + 1) * 2]).pow(2).mean().backward()
        refused = False
    except RuntimeError as e:
        refused = "finish()" in str(e)
    red.finish()
    assert refused
    net.steps_seen += rank                                 # buffers drift per rank; sync_buffers restores rank 0's
    red.sync_buffers()
    assert int(net.steps_seen) == 7
    if rank == 0:
        q.put([o.numpy() for o in out])
    dist.destroy_process_group()


def test_bucket_reducer_matches_single_process_mean():
    world, port = 2, 29000 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=60)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40), torch.nn.Tanh(),
                              torch.nn.Linear(40, 3), torch.nn.Linear(3, 3))
    g = torch.Generator().manual_seed(100)
    x_all = torch.randn(NSTEPS, 2 * world, 12, generator=g)
    for step in range(NSTEPS):
        net.zero_grad(set_to_none=False)
        for p in net.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            p.grad.zero_()
        # mean over ranks of per-rank means == mean over the whole batch (equal shards)
        y = net[:5](x_all[step]) if step == 1 else net(x_all[step])
        y.pow(2).mean().backward()
        ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        assert torch.allclose(torch.from_numpy(got[step]), ref, atol=1e-6), step


def test_local_grads_keeps_the_reducer_surface():
    """LocalGrads (the reducer stand-in when ShardedClipAdamW exchanges the gradients) must offer what the training
    loop, bench.py and GraphedTrainStep call on a GradBucketReducer, and leave local gradients alone."""
    from dtt_amd.dist import GradBucketReducer, LocalGrads
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    m[0].bias.requires_grad = False
    lg = LocalGrads(m)
    for name in ("zero_grad", "finish", "reduce_grads", "pause_hooks", "abort_step", "capturable_collectives"):
        assert callable(getattr(lg, name)) and callable(getattr(GradBucketReducer, name)), name
    assert lg.world == 1 and lg.capturable_collectives() is False and len(lg.params) == 3
    m(torch.randn(5, 4)).sum().backward()
    g = [p.grad.clone() for p in lg.params]
    lg.finish()
    assert all(torch.equal(a, p.grad) for a, p in zip(g, lg.params))
    lg.zero_grad()
    assert all(p.grad is None for p in lg.params)
    lg.reduce_grads(lg.params, g)
    assert all(p.grad is a for a, p in zip(g, lg.params))
