"""The training step recorded into ONE HIP graph (dtt_amd.GraphedTrainStep) must follow the eagerly issued
step: same losses, same weights after several steps with changing inputs and a moving learning rate."""
import pytest
import torch

from oracle.fill import fill_state_dict, synth_sources
from conftest import load_manifest

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(capturable):
    import dtt_amd
    model = dtt_amd.DTTNetModel(g=8, n_sources=2, n_layers=3, n_idp_layers=1, n_fft=64, hop_length=16, n_heads=2,
                                precision="fp32").to(DEV)
    model.load_state_dict(fill_state_dict(load_manifest("tiny_audio"), 1))
    decay = [p for n, p in model.named_parameters() if "bias" not in n]
    no_decay = [p for n, p in model.named_parameters() if "bias" in n]
    opt = dtt_amd.ClipAdamW([{"params": decay, "weight_decay": 1e-2}, {"params": no_decay, "weight_decay": 0.0}],
                            lr=1e-3, max_norm=2.0, capturable=capturable)
    return model, opt, dtt_amd.PITLossWrapper(dtt_amd.DTTNetLoss())


def _batch_fn(inp):
    import dtt_amd
    s1, s2 = inp["s1"], inp["s2"]
    mix = s1 + s2
    spec, phase = dtt_amd.stft_mag_phase(mix, 64, 16)
    sources = []
    for s in (s1, s2):
        sp, ph = dtt_amd.stft_mag_phase(s, 64, 16)
        sources.append({"audio": s, "spectrogram": sp, "phase": ph})
    return {"audio": mix, "spectrogram": spec, "phase": phase, "sources": sources, "audio_length": 800}


def test_graphed_step_follows_eager_step():
    import dtt_amd
    inputs = []
    for i in range(4):
        s1, s2 = synth_sources(2, 800, seed=60 + i)
        inputs.append({"s1": s1.to(DEV), "s2": s2.to(DEV)})
    lrs = [1e-3, 3e-3, 2e-3, 5e-4]

    m_e, o_e, crit = _setup(False)
    w0 = [p.detach().clone() for p in m_e.parameters()]
    losses_e = []
    for inp, lr in zip(inputs, lrs):
        for grp in o_e.param_groups:
            grp["lr"] = lr
        o_e.zero_grad()
        b = _batch_fn(inp)
        b.update(m_e(**b))
        b.update(crit(**b))
        b["loss"].backward()
        o_e.step()
        losses_e.append(float(b["loss"].detach()))

    m_g, o_g, crit = _setup(True)
    step = dtt_amd.GraphedTrainStep(m_g, crit, o_g, inputs[0], _batch_fn, warmup=1)
    losses_g = []
    for inp, lr in zip(inputs, lrs):
        for grp in o_g.param_groups:
            grp["lr"] = lr
        losses_g.append(float(step(inp)))
    # eager work after the capture must not touch graph memory (fresh zero arena): one more eager forward/backward
    b = _batch_fn(inputs[1])
    b.update(m_g(**b))
    b.update(crit(**b))
    b["loss"].backward()
    torch.cuda.synchronize()
    # Same kernels, same order: the first two losses agree to rounding.  From there on Adam's normalised update
    # turns the rounding-level noise of near-zero gradient entries (fp32 atomics in the weight-gradient kernels
    # land in a different order run to run) into lr-sized differences, so later steps are compared by the loss
    # level and by the direction and size of the total weight update.
    for a, b_ in zip(losses_e[:2], losses_g[:2]):
        assert abs(a - b_) <= 1e-5 * abs(a), (losses_e, losses_g)
    for a, b_ in zip(losses_e, losses_g):
        assert abs(a - b_) <= 2e-2 * abs(a), (losses_e, losses_g)
    de = torch.cat([(p.detach() - w).reshape(-1) for p, w in zip(m_e.parameters(), w0)]).double()
    dg = torch.cat([(p.detach() - w).reshape(-1) for p, w in zip(m_g.parameters(), w0)]).double()
    assert float(torch.dot(de, dg) / (de.norm() * dg.norm())) > 0.97
    assert abs(float(dg.norm() / de.norm()) - 1) < 0.05
    assert float(o_g.state_dict()["state"][0]["step"]) == 4.0


def test_pack_plan_matches_plain_steps():
    """ops._PackPlan (derived weight tensors refreshed by one dtt_multi_pack launch per step): an eager
    step with the plan recorded, replayed, and replayed after the parameters moved gives the loss and the
    gradients of the same step without it (to the run-to-run noise of the fp32 atomics)."""
    from dtt_amd import ops
    model, _, crit = _setup(False)
    s1, s2 = synth_sources(2, 800, seed=77)
    inp = {"s1": s1.to(DEV), "s2": s2.to(DEV)}

    def grads():
        for p in model.parameters():
            p.grad = None
        b = _batch_fn(inp)
        b.update(model(**b))
        b.update(crit(**b))
        b["loss"].backward()
        ops.join_packs()
        torch.cuda.synchronize()
        return torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double(), float(b["loss"])

    def close(a, b):
        return float((a - b).norm() / a.norm()) < 1e-5

    ops.weight_plan(False)
    g0, l0 = grads()
    try:
        ops.weight_plan(True)
        g1, l1 = grads()                 # records
        assert len(ops._PLAN.pairs) > 10
        g2, l2 = grads()                 # refreshed by the one pack launch
        assert ops._PLAN.table is not None and ops._PLAN.table[2] > 10
        with torch.no_grad():            # the plan follows the parameters
            for p in model.parameters():
                p.mul_(1.01)
        g3, l3 = grads()
        ops.weight_plan(False)
        g4, l4 = grads()
    finally:
        ops.weight_plan(False)
    assert abs(l0 - l1) <= 1e-6 * abs(l0) and abs(l0 - l2) <= 1e-6 * abs(l0) and abs(l3 - l4) <= 1e-6 * abs(l3)
    assert abs(l3 - l0) > 1e-6 * abs(l0)
    assert close(g0, g1) and close(g0, g2) and close(g4, g3)


def _bench_workload(mode, name="xattn60m"):
    import argparse
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from dtt_amd.dist import GradBucketReducer
    args = argparse.Namespace(batch=None, precision="bf16", graph=mode)
    w = bench.build_workload(name, args, torch.device("cuda", 0), 0, 1, GradBucketReducer)
    assert w["mode"] == ("hip-graph" if mode == "1" else "eager")
    return w


def test_bench_step_keeps_learning_after_a_host_sync():
    """The default bench workload (cfg4: x-attn 60M, B = 16, bf16, H = 512 LSTMs) as a replayed HIP graph, with the
    host synchronising with the stream after five replays -- the pattern of bench.py's warm-up / barrier / timed
    loop.  The non-deep form of lstm_rs_bwd_kernel returned wrong gate gradients from the second replay after such
    a sync on (gradient norm ~1e4 instead of ~15, loss stuck at 4.66 while eager steps reach 3.53: DESIGN finding
    34), because their counters were reset by hipMemsetAsync nodes; a kernel resets them now.  The graph must track eager steps."""
    from dtt_amd import ops
    losses = {}
    for mode in ("0", "1"):
        w = _bench_workload(mode)
        for i in range(5):
            l = w["step"](i)
        torch.cuda.current_stream().synchronize()
        for i in range(5, 24):
            l = w["step"](i)
        torch.cuda.synchronize()
        ops.handoff_check()                # the hand-off kernels' status words stayed 0 across the sync (finding 34)
        losses[mode] = float(l.detach())
        norm = float(w["opt"]._built["norm"])
        assert norm < 200.0, f"gradient norm {norm} in mode {mode}"
        del w
        torch.cuda.empty_cache()
    assert losses["0"] < 4.45 and abs(losses["1"] - losses["0"]) < 0.05, losses


def test_bench_graph_gradients_match_eager_after_a_host_sync():
    """Same workload with the learning rate at zero (weights stay put): the gradients a replay leaves behind -- before
    a host sync, and three replays after one -- against an eager forward + backward over the same batch and weights,
    parameter by parameter.  A loss curve cannot see a gradient that is wrong but bounded; this can."""
    w = _bench_workload("1")
    gs = w["step"].__defaults__[0]
    for grp in w["opt"].param_groups:
        grp["lr"], grp["weight_decay"] = 0.0, 0.0
    names = [k for k, p in w["model"].named_parameters() if p.requires_grad]

    def replay_grads(n):
        for _ in range(n):
            w["step"](0)
        torch.cuda.synchronize()
        return [g.detach().clone() for g in gs.graph_grads]

    from dtt_amd import ops
    before = replay_grads(3)
    ops.handoff_check()                        # no time-out before ...
    torch.cuda.current_stream().synchronize()
    after = replay_grads(3)
    ops.handoff_check()                        # ... or after the host-side sync: pins the fix to the mechanism
    gs.reducer.zero_grad()
    for p in gs.params:
        p.grad = None
    loss = gs._fwd_bwd()                        # same kernel families as recorded (ops.handoff_kernels is restored)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(gs.static_loss)) < 2e-3
    worst = []
    for k, p, gb, ga in zip(names, gs.params, before, after):
        ref = p.grad.float()
        scale = float(ref.norm()) + 1e-6
        eb, ea = float((gb.float() - ref).norm()) / scale, float((ga.float() - ref).norm()) / scale
        # (same weights, same kernels: replay and eager issue differ by the order of split-K atomics only -- measured
        # 5e-7 at worst over all parameters -- so 1e-4 is a sharp bound)
        if max(eb, ea) > 1e-4 and scale > 1e-6:
            worst.append((max(eb, ea), k, eb, ea, scale))
    assert not worst, sorted(worst, reverse=True)[:8]
