"""CPU, world_size 2, gloo: the data-parallel gradient bucket reducer
(dtt_amd/dist.py) averages gradients across ranks exactly like a single-process
run over the concatenated batch, with bucket all-reduces issued from grad hooks."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    x_all = torch.randn(2, 2 * world, 12, generator=g)
    out = []
    for step in range(2):
        red.zero_grad()
        x = x_all[step, rank * 2:(rank + 1) * 2]
        y = net[:5](x) if step == 1 else net(x)
        y.pow(2).mean().backward()
        red.finish()
        out.append(torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone())
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
    x_all = torch.randn(2, 2 * world, 12, generator=g)
    for step in range(2):
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
