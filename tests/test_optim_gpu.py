"""dtt_clip_adamw (two launches) against torch's clip_grad_norm_ + AdamW: the step the reference's
trainer takes right after the hot path (base_trainer.py:478-490, trainer.py:61-64, train.yaml:20-23,59)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("max_norm", [2.0, 1e6, None])
def test_clip_adamw_matches_torch(max_norm):
    import dtt_amd
    g = torch.Generator().manual_seed(11)
    shapes = [(64, 32, 3, 3), (32,), (1024, 257), (7,), (3, 5, 1), (70001,), (1,)]
    ref = [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    our = [p.detach().clone().requires_grad_(True) for p in ref]
    def groups(ps):
        return [{"params": [p for p in ps if p.dim() > 1], "weight_decay": 1e-2},
                {"params": [p for p in ps if p.dim() <= 1], "weight_decay": 0.0, "lr": 3e-3}]
    o_ref = torch.optim.AdamW(groups(ref), lr=1e-3)
    o_our = dtt_amd.ClipAdamW(groups(our), lr=1e-3, max_norm=max_norm)
    sched_ref = torch.optim.lr_scheduler.OneCycleLR(o_ref, max_lr=[2e-3, 4e-3], total_steps=8)
    sched_our = torch.optim.lr_scheduler.OneCycleLR(o_our, max_lr=[2e-3, 4e-3], total_steps=8)
    for it in range(6):
        scale = 10.0 if it % 2 else 0.01
        for i, (a, b) in enumerate(zip(ref, our)):
            gr = torch.randn(a.shape, generator=g).to(DEV) * scale
            skip = it == 3 and i == 2                         # a parameter without a gradient this step
            a.grad = None if skip else gr.clone()
            b.grad = None if skip else gr.clone()
        norm_ref = None
        if max_norm is not None:
            norm_ref = torch.nn.utils.clip_grad_norm_([p for p in ref], max_norm)
        o_ref.step()
        norm = o_our.step()
        sched_ref.step(); sched_our.step()
        if max_norm is not None:
            assert abs(float(norm) - float(norm_ref)) <= 1e-5 * float(norm_ref)
        if it == 3:
            continue                                          # torch keeps a per-parameter step count: see below
    for a, b in zip(ref, our):
        # parameter 2 skipped one step: torch's bias correction for it lags by one step from then on, ours
        # uses the global step (documented); everything else agrees to fp32 rounding
        tol = 2e-3 if a.shape == (1024, 257) else 2e-6
        assert float((a - b).detach().abs().max()) <= tol * float(a.detach().abs().max()), a.shape
    st = o_our.state_dict()
    assert set(st["state"][0].keys()) == {"exp_avg", "exp_avg_sq", "step"} and float(st["state"][0]["step"]) == 6.0


def test_clip_adamw_rejects_bad_layouts():
    import dtt_amd
    from dtt_amd._lib import DttError
    p = torch.randn(8, 8, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    o = dtt_amd.ClipAdamW([p])
    p.grad = torch.zeros_like(p)
    with pytest.raises(DttError):
        o.step()
    with pytest.raises(ValueError):
        dtt_amd.ClipAdamW([torch.zeros(2, device=DEV, requires_grad=True)], betas=(1.0, 0.9))


def _mk(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in shapes], g


def test_clip_adamw_resume_from_state_dict():
    """N steps, save, load into (a) a fresh and (b) an already-stepped ClipAdamW, continue: both must follow
    torch.optim.AdamW + clip_grad_norm_ run without interruption (the loaded moments and step count are used)."""
    import dtt_amd
    shapes = [(33, 17), (129,), (5, 4, 3)]
    ref, g = _mk(shapes, 3)
    ours = [[p.detach().clone().requires_grad_(True) for p in ref] for _ in range(3)]
    o_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2)
    o = [dtt_amd.ClipAdamW(ps, lr=1e-2, weight_decay=1e-2, max_norm=1.0) for ps in ours]

    def feed(which):
        grads = [torch.randn(s, generator=g).to(DEV) for s in shapes]
        for ps in [ref] + [ours[i] for i in which]:
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        o_ref.step()
        for i in which:
            o[i].step()

    for _ in range(3):
        feed([0])
    sd = o[0].state_dict()
    # (a) fresh optimizer, parameters copied
    for a, b in zip(ours[0], ours[1]):
        b.data.copy_(a.data)
    o[1].load_state_dict(sd)
    # (b) an optimizer that has already stepped on other values: its live moments must be replaced
    for p in ours[2]:
        p.grad = torch.ones_like(p)
    o[2].step()
    for a, b in zip(ours[0], ours[2]):
        b.data.copy_(a.data)
    o[2].load_state_dict(sd)
    for _ in range(3):
        feed([0, 1, 2])
    for k in range(3):
        for a, b in zip(ref, ours[k]):
            assert float((a - b).detach().abs().max()) <= 3e-6 * float(a.detach().abs().max()), (k, a.shape)
    assert float(o[2].state_dict()["state"][0]["step"]) == 6.0


def test_clip_adamw_capturable_graph_replay_and_grad_scale():
    """capturable=True: step() recorded in a HIP graph and replayed with new gradients and a new learning rate
    follows the eager optimizer; grad_scale = 1/2 on doubled gradients gives the same update."""
    import dtt_amd
    shapes = [(64, 48), (48,), (3, 3, 8, 8)]
    eager, g = _mk(shapes, 5)
    cap = [p.detach().clone().requires_grad_(True) for p in eager]
    o_e = dtt_amd.ClipAdamW(eager, lr=1e-2, weight_decay=1e-3, max_norm=0.5)
    o_c = dtt_amd.ClipAdamW(cap, lr=1e-2, weight_decay=1e-3, max_norm=0.5, capturable=True, grad_scale=0.5)
    static = [torch.zeros(s, device=DEV) for s in shapes]
    for p, s_ in zip(cap, static):
        p.grad = s_
    lrs = [1e-2, 2e-2, 5e-3, 1e-3]
    o_c.prepare()
    graph = torch.cuda.CUDAGraph()
    for it, lr in enumerate(lrs):
        grads = [torch.randn(s, generator=g).to(DEV) for s in shapes]
        for p, gr in zip(eager, grads):
            p.grad = gr.clone()
        for grp in o_e.param_groups + o_c.param_groups:
            grp["lr"] = lr
        o_e.step()
        for s_, gr in zip(static, grads):
            s_.copy_(2.0 * gr)
        if it == 0:
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                o_c.step()
            o_c.note_recorded_step()            # recording executes nothing
        o_c.refresh_hyper()                     # table upload (current lr), stream-ordered ahead of the replay
        graph.replay()
        o_c.note_replayed_step()
        torch.cuda.synchronize()
    for a, b in zip(eager, cap):
        assert float((a - b).detach().abs().max()) <= 3e-6 * float(a.detach().abs().max()), a.shape
    assert float(o_c.state_dict()["state"][0]["step"]) == float(len(lrs))
