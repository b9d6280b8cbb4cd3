"""What the pack plan copies per step, by access pattern: python tools/plan_census.py [workload]"""
import argparse, collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/speech-separation_amd")
import bench
from dtt_amd import ops
from dtt_amd.dist import GradBucketReducer
args = argparse.Namespace(batch=None, precision="bf16", graph="0")
w = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "xattn60m", args, torch.device("cuda", 0), 0, 1, GradBucketReducer)
ops.weight_plan(True)
for i in range(2):
    w["step"](i)
torch.cuda.synchronize()
hist = collections.Counter()
cnt = collections.Counter()
for d, v in ops._PLAN.pairs:
    dims = ops._merge_dims(v.shape, v.stride(), d.stride())
    n, ss, ds = dims[-1]
    key = (len(dims), "inner n>=%d" % (8 if n >= 8 else n), "ss=%s" % (ss if ss < 16 else "big"), "ds=%s" % (ds if ds < 16 else "big"), str(v.dtype)[6:], str(d.dtype)[6:])
    hist[key] += v.numel()
    cnt[key] += 1
tot = sum(hist.values())
print("pairs", len(ops._PLAN.pairs), "elements", tot)
for k, n in hist.most_common():
    print("%10d %5.1f%% %4d  %s" % (n, 100.0 * n / tot, cnt[k], k))
big = sorted(ops._PLAN.pairs, key=lambda p: -p[1].numel())[:12]
for d, v in big:
    print(v.numel(), tuple(v.shape), v.stride(), "->", d.stride(), v.dtype, d.dtype)
# the refresh launch by itself
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.prepare_packs()
e0.record()
for _ in range(20):
    ops.prepare_packs()
e1.record()
torch.cuda.synchronize()
print("prepare_packs: %.1f us per call (incl. the closures)" % (e0.elapsed_time(e1) / 20 * 1e3), "chunks", ops._PLAN.table[2])
# each access pattern by itself
import ctypes as C
from dtt_amd._lib import lib, check, ptr, stream_ptr
groups = collections.defaultdict(list)
for d, v in ops._PLAN.pairs:
    dims = ops._merge_dims(v.shape, v.stride(), d.stride())
    n, ss, ds = dims[-1]
    groups[(len(dims), "ss=%s" % (ss if ss < 16 else "big"), str(d.dtype)[6:])].append((d, v))
for key, prs in groups.items():
    pl = ops._PackPlan()
    pl.pairs = prs
    pl.build()
    descs, chunks, n = pl.table
    for _ in range(3):
        lib().dtt_multi_pack(ptr(descs), ptr(chunks), n, stream_ptr())
    e0.record()
    for _ in range(10):
        lib().dtt_multi_pack(ptr(descs), ptr(chunks), n, stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    el = sum(v.numel() for _, v in prs)
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("%-34s pairs %4d elements %10d chunks %6d  %7.1f us  %6.0f GB/s" % (key, len(prs), el, n, us, el * 6 / us / 1e3))
pat = collections.Counter()
for d, v in ops._PLAN.pairs:
    dims = ops._merge_dims(v.shape, v.stride(), d.stride())
    if len(dims) == 2 and dims[-1][1] >= 16:
        pat[(tuple(v.shape), tuple(v.stride()), tuple(d.stride()), v.storage_offset() % 9)] += 1
for k, n in sorted(pat.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][0]).numel())[:40]:
    print(n, torch.Size(k[0]).numel() * n, k)
