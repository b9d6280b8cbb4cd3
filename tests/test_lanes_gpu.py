"""Side streams of the eagerly issued step (ops.wgrad_lane: weight gradients beside the data-gradient chain;
ops.video_lane: the lip encoder beside the audio encoder): same losses, gradients and updated parameters as the
one-stream step, step after step (the joins order every consumer), and a graph owner can hand the optimizer back its
static gradients after eager steps (GraphedTrainStep.rearm).  Reference step: src/trainer/trainer.py:37-72."""
import copy

import pytest
import torch

from oracle.fill import synth_video
from helpers import grad_close, make_batch, rel_err, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(seed, precision):
    import dtt_amd
    torch.manual_seed(seed)
    return dtt_amd.DTTNetVisionModel(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2,
                                     video_channels=512, fusion_method="cross_attention_positional",
                                     precision=precision).to(DEV)


def _steps(model, lanes, nsteps, precision):
    import dtt_amd
    from dtt_amd import ops
    keep = (ops._WGRAD_LANE[0], ops._VIDEO_LANE[0])
    ops.wgrad_lane(lanes)
    ops.video_lane(lanes)
    try:
        model.train()
        opt = dtt_amd.ClipAdamW(model.parameters(), lr=1e-3, max_norm=2.0)
        crit = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())
        video = synth_video(2, 4, seed=77).to(DEV)
        losses, grads = [], None
        for i in range(nsteps):
            b = to_dev(make_batch(2, 800, 64, 16, seed=50 + i), DEV)
            for p in model.parameters():
                p.grad = None
            b.update(model(video=video, **b))
            b.update(crit(**b))
            b["loss"].backward()
            if i == 0:                     # before any update: later steps inherit the run-to-run noise of the atomics
                grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            opt.step()
            losses.append(float(b["loss"].detach()))
        torch.cuda.synchronize()
        return losses, grads, {n: p.detach().clone() for n, p in model.named_parameters()}
    finally:
        ops.wgrad_lane(keep[0])
        ops.video_lane(keep[1])


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_side_streams_change_nothing_but_the_schedule(precision, tol):
    base = _model(3, precision)
    other = copy.deepcopy(base)
    again = copy.deepcopy(base)
    l0, g0, p0 = _steps(base, False, 3, precision)
    l1, g1, p1 = _steps(other, True, 3, precision)
    l2, g2, p2 = _steps(again, False, 3, precision)        # the one-stream step against itself: the noise floor
    # split-K weight gradients add their partial sums with fp32 atomics (order varies run to run): close, not identical
    for a, b in zip(l0, l1):
        assert abs(a - b) <= tol * abs(a), (l0, l1)
    assert set(g0) == set(g1) and len(g0) > 100
    # (grad_close: gradients that are mathematically zero -- a bias ahead of a normalisation -- hold rounding noise only)
    atol = 1e-8
    # first-step gradients (same weights, same forward): the side streams change the order of split-K atomics and
    # nothing else -- measured 2e-7 in both precisions (tools/lanes_grad_agreement.py), so a bound of 2e-5 is a sharp
    # detector of a gradient read before its side-stream kernel wrote it
    bad = [k for k in g0 if not grad_close(g1[k], g0[k], max(2e-5, 20 * rel_err(g2[k], g0[k])), atol=atol)]
    assert not bad, [(k, rel_err(g1[k], g0[k]), rel_err(g2[k], g0[k]), float(g0[k].abs().max())) for k in bad[:5]]
    # three updates later: within a few times the noise floor.  (The one-stream step repeats itself bit for bit, so
    # the floor can be 0; a different order of the split-K atomics is a rounding-level change of the first update that
    # bf16 activations re-round into percent-level changes of small gradients deep in the lip encoder, which Adam's
    # normalisation passes on undamped: seen 3-4 % on one BatchNorm bias, hence 3 x the layer tolerance.)
    # Elements whose gradient is rounding noise (mathematically zero: biases ahead of a normalisation, the key third
    # of in_proj_bias under a softmax) walk wherever Adam's normalisation takes that noise: only elements with a
    # first-step gradient of at least 1 % of their tensor's largest are compared.
    # fp32 only: in bf16 a rounding-level change of the first update is re-rounded by every bf16 activation into
    # percent-level changes of the following gradients, and three normalised steps later single tensors sit 3-10 %
    # apart (seen on a BatchNorm bias, on in_proj_bias, on a LayerNorm bias in turn): nothing a bound can be put on.
    # The losses of the three steps above are compared in both precisions.
    checked = 0
    for k in (p0 if precision == "fp32" else ()):
        if k not in g0:
            continue
        m = g0[k].abs() >= 1e-2 * g0[k].abs().max().clamp(min=1e-30)
        if int(m.sum()) == 0:
            continue
        e, floor = rel_err(p1[k][m], p0[k][m]), rel_err(p2[k][m], p0[k][m])
        assert e < max(3 * tol, 20 * floor), (k, e, floor)
        checked += 1
    assert checked > 100 or precision != "fp32"


def test_side_streams_survive_a_consumer_inside_backward():
    """A post-accumulate hook reads .grad while the backward pass is still running: after join_wgrad_lane() (what
    GradBucketReducer's bucket launch calls) the value is final."""
    from dtt_amd import ops
    import dtt_amd
    model = _model(4, "fp32")
    seen = {}

    def hook(p):
        ops.join_wgrad_lane()
        seen[id(p)] = p.grad.detach().clone()
    hs = [p.register_post_accumulate_grad_hook(hook) for p in model.parameters() if p.requires_grad]
    crit = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())
    b = to_dev(make_batch(2, 800, 64, 16, seed=60), DEV)
    b.update(model(video=synth_video(2, 4, seed=78).to(DEV), **b))
    b.update(crit(**b))
    b["loss"].backward()
    torch.cuda.synchronize()
    for h in hs:
        h.remove()
    n = 0
    for p in model.parameters():
        if p.grad is not None:
            assert torch.equal(seen[id(p)], p.grad)
            n += 1
    assert n > 100


def test_graph_step_rearms_after_eager_steps():
    """bench.py --graph auto times eager steps between graph replays: rearm() must hand the recorded optimizer step its
    static gradients back, or every other replay would update from a stale eager gradient."""
    import dtt_amd
    from dtt_amd import ops
    from oracle.fill import synth_sources
    from test_graph_gpu import _setup, _batch_fn
    pool = []
    for i in range(2):
        s1, s2 = synth_sources(2, 800, seed=80 + i)
        pool.append({"s1": s1.to(DEV), "s2": s2.to(DEV)})

    def run(with_graph):
        m, opt, crit = _setup(with_graph)

        def eager(i):
            b = _batch_fn(pool[i % 2])
            for p in m.parameters():
                p.grad = None
            b.update(m(**b))
            b.update(crit(**b))
            b["loss"].backward()
            opt.step()
            return float(b["loss"].detach())
        if not with_graph:
            return [eager(i) for i in range(6)]
        gs = dtt_amd.GraphedTrainStep(m, crit, opt, pool[0], _batch_fn, warmup=1)
        out = [float(gs(pool[0])), float(gs(pool[1]))]
        out += [eager(0), eager(1)]
        gs.rearm()
        for i in range(2):                                        # two replays: the optimizer alternates two tables
            out.append(float(gs(pool[i])))
            seen = float(opt._built["norm"])                      # the norm the recorded clip + AdamW computed ...
            have = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gs.graph_grads if g is not None)))
            assert abs(seen - have) < 1e-4 * have, (i, seen, have)    # ... is the norm of the graph's own gradients
        return out
    try:
        want = run(False)
        got = run(True)
    finally:
        ops.reset_zero_arena()
    assert want[0] > want[-1]                                    # the steps train
    for i, (a, b) in enumerate(zip(got, want)):                  # (later steps drift like any two runs: test_graph_gpu.py)
        assert abs(a - b) < (1e-5 if i < 2 else 2e-2) * abs(b), (got, want)


def test_side_streams_stand_aside_for_accumulation_and_parameter_slices():
    """Gradients that something reads on the launch stream INSIDE the backward pass stay off the side stream
    (ops._lane_ok): a second backward into existing .grad tensors (AccumulateGrad adds), and weights that are slices
    of a parameter (the q / k / v thirds of in_proj_weight: SliceBackward copies).  Two accumulated micro-batches with
    side streams on equal the same with side streams off."""
    import dtt_amd
    from dtt_amd import ops
    assert not ops._lane_ok(torch.nn.Parameter(torch.zeros(4, 4, device=DEV))[:2])          # a slice: has a grad_fn
    p = torch.nn.Parameter(torch.zeros(4, device=DEV))
    assert ops._lane_ok(p, None)
    p.grad = torch.zeros_like(p)
    assert not ops._lane_ok(p)

    def run(lanes):
        model = _model(9, "fp32")
        keep = (ops._WGRAD_LANE[0], ops._VIDEO_LANE[0])
        ops.wgrad_lane(lanes)
        ops.video_lane(lanes)
        try:
            crit = dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())
            video = synth_video(2, 4, seed=91).to(DEV)
            for i in range(2):                                    # no zeroing in between: the second pass accumulates
                b = to_dev(make_batch(2, 800, 64, 16, seed=70 + i), DEV)
                b.update(model(video=video, **b))
                b.update(crit(**b))
                b["loss"].backward()
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        finally:
            ops.wgrad_lane(keep[0])
            ops.video_lane(keep[1])
    g0, g1, g2 = run(False), run(True), run(False)
    bad = [k for k in g0 if not grad_close(g1[k], g0[k], max(2e-3, 20 * rel_err(g2[k], g0[k])), atol=2e-6)]
    assert not bad, [(k, rel_err(g1[k], g0[k]), rel_err(g2[k], g0[k])) for k in bad[:5]]
    assert "fusion.cross_attn.in_proj_weight" in g0


def test_a_backward_pass_that_raised_does_not_leave_the_next_one_unjoined():
    """The join is queued once per autograd task (by task id): after a pass that raised before its final callbacks ran,
    the next pass queues its own and ends joined."""
    from dtt_amd import ops
    keep = ops._WGRAD_LANE[0]
    ops.wgrad_lane(True)
    try:
        w = torch.nn.Parameter(torch.randn(64, 32, device=DEV))
        x = torch.randn(256, 32, device=DEV)

        class Boom(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return t.clone()

            @staticmethod
            def backward(ctx, g):
                raise RuntimeError("boom")
        y = ops.Linear.apply(Boom.apply(x.requires_grad_()), w, None, False)
        with pytest.raises(RuntimeError):
            y.sum().backward()                  # Linear's backward forked, Boom raised: no final callback ran
        ops.join_wgrad_lane()                   # (what the next optimizer step / loss wrapper does anyway)
        w.grad = None
        y = ops.Linear.apply(x.detach(), w, None, False)
        y.sum().backward()
        ln = ops._LANES[torch.cuda.current_device()]
        assert not ln.pending                   # this pass queued its own join and it ran
        torch.cuda.synchronize()
        want = torch.ones(256, 64, device=DEV).t() @ x.detach()
        assert rel_err(w.grad, want) < 1e-4
        # and without the explicit join in between
        y = ops.Linear.apply(Boom.apply(x), w, None, False)
        with pytest.raises(RuntimeError):
            y.sum().backward()
        w.grad = None
        ops.Linear.apply(x.detach(), w, None, False).sum().backward()
        assert not ln.pending
        torch.cuda.synchronize()
        assert rel_err(w.grad, want) < 1e-4
    finally:
        ops.join_wgrad_lane()
        ops.wgrad_lane(keep)


@pytest.mark.parametrize("workload,nparams", [("xattn60m", 300), ("audio15m", 150), ("film25m", 150)])
def test_side_streams_at_the_bench_configuration(workload, nparams):
    """The bench workloads (cfg4: x-attn 60M with the lip encoder, B = 16; cfg2: audio-only 15M, B = 64; cfg3: FiLM 25M
    with the frozen ShuffleNetV2 front end, B = 64; bf16), eagerly issued as bench.py times them: gradients of one step
    with the side streams on against the one-stream step on the same weights, every parameter, to the order-of-atomics
    floor.  At these sizes the side streams really run beside the launch stream (5 ms of overlap per cfg4 step), which
    the tiny model above cannot promise."""
    from dtt_amd import ops
    from test_graph_gpu import _bench_workload
    w = _bench_workload("0", workload)
    model = w["model"]
    for grp in w["opt"].param_groups:
        grp["lr"], grp["weight_decay"] = 0.0, 0.0              # the weights stay put

    def grads(lanes):
        keep = (ops._WGRAD_LANE[0], ops._VIDEO_LANE[0])
        ops.wgrad_lane(lanes)
        ops.video_lane(lanes)
        try:
            w["eager_step"](0)
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        finally:
            ops.wgrad_lane(keep[0])
            ops.video_lane(keep[1])
    try:
        grads(False)                                           # (records the pack plan, warms the allocator)
        g0, g1, g2, g3 = grads(False), grads(True), grads(False), grads(True)
    finally:
        ops.weight_plan(False)
        ops.reset_zero_arena()
    assert len(g0) > nparams
    bad = []
    for k in g0:
        floor = rel_err(g2[k], g0[k])
        for g in (g1, g3):
            if not grad_close(g[k], g0[k], max(1e-4, 20 * floor), atol=1e-8):
                bad.append((k, rel_err(g[k], g0[k]), floor, float(g0[k].abs().max())))
    assert not bad, bad[:6]


def test_pack_plan_at_the_bench_configuration():
    """Every derived weight tensor of the cfg4 step from the one dtt_multi_pack launch (contiguous casts, tap-major and
    backwards-walked conv packs, LSTM transposes, the slices of in_proj_weight) against the per-use packs: gradients
    of a step on the same weights, before and after the weights moved (the plan follows the parameters), every
    parameter at the atomics floor."""
    from dtt_amd import ops
    from test_graph_gpu import _bench_workload
    w = _bench_workload("0")
    model = w["model"]
    for grp in w["opt"].param_groups:
        grp["lr"], grp["weight_decay"] = 0.0, 0.0
    keep = (ops._WGRAD_LANE[0], ops._VIDEO_LANE[0])
    ops.wgrad_lane(False)
    ops.video_lane(False)

    def grads():
        w["eager_step"](0)
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    def agree(a, b, floor):
        return [(k, rel_err(a[k], b[k])) for k in b if not grad_close(a[k], b[k], max(1e-4, 20 * rel_err(floor[k], b[k])), atol=1e-8)]
    try:
        assert ops._PLAN.enabled
        grads()                                                # records the plan
        assert len(ops._PLAN.pairs) > 150
        on1 = grads()                                          # refreshed by the one launch
        orig = [p.detach().clone() for p in model.parameters()]
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.02)
        on2 = grads()
        ops.weight_plan(False)
        off2, off2b = grads(), grads()
        with torch.no_grad():
            for p, o in zip(model.parameters(), orig):
                p.copy_(o)
        off1, off1b = grads(), grads()
        bad2, bad1 = agree(on2, off2, off2b), agree(on1, off1, off1b)
        assert not bad2, ("after the weights moved", bad2[:6])
        assert not bad1, ("on the first weights", bad1[:6])
        moved = [k for k in off1 if rel_err(off2[k], off1[k]) > 1e-3]
        assert len(moved) > 200                                # (the two weight sets really give different gradients)
    finally:
        ops.weight_plan(False)
        ops.reset_zero_arena()
        ops.wgrad_lane(keep[0])
        ops.video_lane(keep[1])
