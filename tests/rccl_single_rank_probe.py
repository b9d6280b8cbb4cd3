"""RCCL path of GradBucketReducer on one GPU: a one-rank "nccl" group, the reducer told the world is 2 so that its
hooks, asynchronous bucket all-reduces (ReduceOp.AVG inside the collective) and waits all run; an average over one
rank is the identity, so the gradients must come out unchanged and one collective per bucket must have been issued.
Second leg: the gradients of a replayed HIP graph exchanged through reduce_grads()."""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-separation_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29000 + os.getpid() % 2000))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import dtt_amd
from dtt_amd import dist as ddist
torch.manual_seed(1)
model = dtt_amd.DTTNetModel(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2, precision="fp32").cuda()
real = dist.get_world_size
dist.get_world_size = lambda group=None: 2
red = ddist.GradBucketReducer(model, bucket_bytes=64 << 10)
dist.get_world_size = real
print("buckets", len(red.buckets), "host_staged", red._host_staged, flush=True)
calls = []
real_ar = dist.all_reduce
def counting_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
    calls.append(op)
    return real_ar(t, op=op, group=group, async_op=async_op)
dist.all_reduce = counting_all_reduce
x = torch.randn(2, 800, device="cuda")
def run(reducer):
    spec, phase = dtt_amd.stft_mag_phase(x, 64, 16)
    out = model(spectrogram=spec, phase=phase, audio_length=800)
    loss = sum(p["audio"].pow(2).mean() for p in out["preds"])
    if reducer: reducer.zero_grad()
    else: model.zero_grad()
    loss.backward()
    if reducer: reducer.finish()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in model.parameters()]
g_red = run(red)
red.remove()
g_ref = run(None)
worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(g_red, g_ref))
print("worst rel deviation from the plain gradient:", worst, "collectives", len(calls), flush=True)
assert worst < 1e-5
assert len(calls) == len(red.buckets) and all(op == dist.ReduceOp.AVG for op in calls)
# ---- second leg: forward + backward as a replayed HIP graph, gradients exchanged behind the replay ----
dist.get_world_size = lambda group=None: 2
red2 = ddist.GradBucketReducer(model, bucket_bytes=64 << 10)
dist.get_world_size = real
crit = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())
opt = dtt_amd.ClipAdamW(model.parameters(), lr=1e-3, max_norm=2.0)
s1, s2 = torch.randn(2, 800, device="cuda") * 0.1, torch.randn(2, 800, device="cuda") * 0.1
def batch_fn(inp):
    mix = inp["s1"] + inp["s2"]
    spec, phase = dtt_amd.stft_mag_phase(mix, 64, 16)
    srcs = []
    for s in (inp["s1"], inp["s2"]):
        sp, ph = dtt_amd.stft_mag_phase(s, 64, 16)
        srcs.append({"audio": s, "spectrogram": sp, "phase": ph})
    return {"audio": mix, "spectrogram": spec, "phase": phase, "sources": srcs, "audio_length": 800}
w_before = [p.detach().clone() for p in model.parameters()]
del calls[:]
step = dtt_amd.GraphedTrainStep(model, crit, opt, {"s1": s1, "s2": s2}, batch_fn, reducer=red2, warmup=1)
n_warm = len(calls)
l1 = float(step({"s1": s1, "s2": s2}))
l2 = float(step({"s1": s1, "s2": s2}))
torch.cuda.synchronize()
print("graph leg: losses", l1, l2, "collectives per step", (len(calls) - n_warm) // 2, flush=True)
assert len(calls) - n_warm == 2 * len(red2.buckets)
assert l2 < l1 and all(torch.isfinite(p).all() for p in model.parameters())
assert any(float((p.detach() - w).abs().max()) > 0 for p, w in zip(model.parameters(), w_before))
# ---- third leg: the sharded step's collectives on RCCL (reduce_scatter_tensor AVG, all_reduce, all_gather_into_tensor)
torch.manual_seed(1)
m2 = dtt_amd.DTTNetModel(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2, precision="fp32").cuda()
m3 = dtt_amd.DTTNetModel(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2, precision="fp32").cuda()
m3.load_state_dict(m2.state_dict())
sh = ddist.ShardedClipAdamW(m2, lr=1e-3, weight_decay=1e-5, max_norm=0.5, force_collectives=True)
rep = dtt_amd.ClipAdamW(m3.parameters(), lr=1e-3, weight_decay=1e-5, max_norm=0.5)
w0 = [p.detach().clone() for p in m3.parameters()]
for mm, oo in ((m2, sh), (m3, rep)):
    spec, phase = dtt_amd.stft_mag_phase(x, 64, 16)
    out = mm(spectrogram=spec, phase=phase, audio_length=800)
    sum(p["audio"].pow(2).mean() for p in out["preds"]).backward()
    nrm = float(oo.step())
    print("norm", nrm, flush=True)
torch.cuda.synchronize()
# What CAN fail: the first moments are linear in the clipped gradient (reduce-scatter, segment offsets, clip
# coefficient); the parameter step where |g| >> eps is -lr * sign(g) (the update and the all-gather); everything with
# a gradient must have moved.  (A bound of ~lr on the parameters alone would pass with no update at all.)
shard = sh.state_shard()
coef = min(1.0, 0.5 / (nrm + 1e-6))                          # clip coefficient of clip_grad_norm_ (max_norm 0.5)
worst_m, worst_d, n_big = 0.0, 0.0, 0
for (k, p2), p3, w in zip(m2.named_parameters(), m3.parameters(), w0):
    off, m, v = shard[k]
    rm = rep.state[p3]["exp_avg"].reshape(-1)
    assert m.numel() == rm.numel(), k                      # one rank: the shard is everything
    if float(rm.norm()) > 0:
        worst_m = max(worst_m, float((m.reshape(-1) - rm).norm() / rm.norm()))
    gc = p3.grad * coef
    big = gc.abs() > 1e-6                                  # |gc| >> eps: the first AdamW step is -lr gc / (|gc| + eps)
    if bool(big.any()):
        n_big += 1
        want = -1e-3 * gc[big] / (gc[big].abs() + 1e-8)
        d2, d3 = (p2.detach() - w)[big], (p3.detach() - w)[big]
        assert float((d3 - want).abs().max()) < 2e-5, k    # the replicated step is AdamW's
        worst_d = max(worst_d, float((d2 - want).abs().max()))
    if float(p3.grad.abs().max()) > 0:
        assert float((p2.detach() - w).abs().max()) > 0, k     # the sharded step moved the tensor at all
print("sharded (RCCL RS/AG, one rank) vs replicated step: worst relative first-moment difference", worst_m,
      "worst parameter-step error where |gc| > 1e-6:", worst_d, "over", n_big, "tensors", flush=True)
assert worst_m < 1e-4 and worst_d < 2e-5 and n_big > 50
# ---- fourth leg: the bucket all-reduces recorded INTO the HIP graph (GraphedTrainStep overlap mode: what
#      bench.py --gpus N runs on RCCL): forward + backward + collectives + clip/AdamW as one launch
torch.manual_seed(2)
m4 = dtt_amd.DTTNetModel(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2, precision="fp32").cuda()
dist.get_world_size = lambda group=None: 2
red4 = ddist.GradBucketReducer(m4, bucket_bytes=64 << 10)
dist.get_world_size = real
opt4 = dtt_amd.ClipAdamW(m4.parameters(), lr=0.0, weight_decay=0.0, max_norm=2.0, capturable=True)
del calls[:]
step4 = dtt_amd.GraphedTrainStep(m4, crit, opt4, {"s1": s1, "s2": s2}, batch_fn, reducer=red4, warmup=1)
n_rec = len(calls)
assert step4.overlap, "the collectives were not recorded into the graph (see stderr)"
assert n_rec == 2 * len(red4.buckets), (n_rec, len(red4.buckets))        # one warm-up step + the recording
la = float(step4({"s1": s1, "s2": s2}))
torch.cuda.synchronize()
lb = float(step4({"s1": s1, "s2": s2}))
torch.cuda.synchronize()
assert len(calls) == n_rec                                             # replays call nothing from Python
g_graph = [p.grad.detach().clone() for p in m4.parameters()]
assert all(p.grad.data_ptr() == v.data_ptr() for (_, ps, vs) in red4.buckets for p, v in zip(ps, vs))   # bucket views
red4.remove()
for p in m4.parameters():
    p.grad = None
b4 = batch_fn({"s1": s1, "s2": s2})
b4.update(m4(**b4)); b4.update(crit(**b4)); b4["loss"].backward()
torch.cuda.synchronize()
worst4 = max(float((a - p.grad).norm() / (p.grad.norm() + 1e-12)) for a, p in zip(g_graph, m4.parameters()) if float(p.grad.norm()) > 1e-6)
print("graph with recorded collectives: losses", la, lb, "eager", float(b4["loss"]), "worst gradient deviation", worst4, flush=True)
assert abs(la - float(b4["loss"])) < 1e-5 * abs(la) and la == lb and worst4 < 1e-4
# ---- fifth leg: RCCL all-reduces in flight BESIDE the hand-off LSTM kernels (lstm_rs.hip needs every workgroup of a
#      launch co-resident, one per CU; an RCCL kernel holds CUs of the same GPU while backward runs).  The launches
#      may stall until the collective's workgroups leave, but must not time out: the status words stay 0.
from dtt_amd import ops
from dtt_amd.model import IDPModule
idp = IDPModule(2, 512).cuda()
xi = (torch.randn(4, 33, 16, 512, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_(True)
big = torch.randn(16 << 20, device="cuda")
ops.handoff_check()
y0 = idp(xi).detach().clone()
works = []
for rep in range(6):
    works.append(real_ar(big, op=dist.ReduceOp.AVG, async_op=True))
    yi = idp(xi)
    works.append(real_ar(big, op=dist.ReduceOp.AVG, async_op=True))
    yi.float().sum().backward()
for wk in works:
    wk.wait()
torch.cuda.synchronize()
ops.handoff_check()                       # raises DttError on a timed-out hand-off
assert torch.equal(yi.detach(), y0), "a hand-off LSTM launch beside an RCCL all-reduce returned different hidden states"
print("hand-off LSTM kernels beside 12 RCCL all-reduces of 64 MB: status words 0, outputs bit-identical", flush=True)
dist.destroy_process_group()
print("rccl single-rank probe ok")
