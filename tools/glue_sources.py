#!/usr/bin/env python3
"""Where the ATen (non-library) launches of one eager training step come from.

    python3 tools/glue_sources.py [workload] > gpurun_out/glue.txt

Runs the bench workload eagerly under a TorchDispatchMode (backward on the calling thread, so the mode sees it too)
and prints, per ATen operator and issuing line inside dtt_amd/, the calls per step and the bytes they write.  The
per-kernel tables under profiles/ say HOW MUCH time torch glue takes (copies, casts, adds, cats, fills); this says
WHICH LINE asked for it.
"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.as_strided", "aten.t.", "aten.transpose", "aten.permute",
        "aten.slice", "aten.select", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.reshape",
        "aten.empty", "aten.unbind", "aten.split", "aten.narrow", "aten.unflatten", "aten.flatten", "aten.chunk",
        "aten.new_empty", "aten.empty_like", "aten.lift_fresh", "aten._local_scalar", "aten.is_", "aten.sym_",
        "aten.stride", "aten.size", "aten.movedim", "aten.diagonal", "aten.unfold", "aten.view_as")


class Tally(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.defaultdict(lambda: [0, 0])
        self.h2d = collections.Counter()     # ops that read a CPU tensor and write a GPU one: host memory inside the step

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        flat = list(args) + list((kwargs or {}).values())
        flat = [a for x in flat for a in (x if isinstance(x, (list, tuple)) else [x])]
        o0 = out[0] if isinstance(out, (tuple, list)) and out else out
        if (isinstance(o0, torch.Tensor) and o0.is_cuda and any(isinstance(a, torch.Tensor) and not a.is_cuda and a.dim() > 0
                                                                 for a in flat)):
            fr = [f for f in traceback.extract_stack() if "dtt_amd/" in f.filename or f.filename.endswith("bench.py")]
            self.h2d[(name, " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in reversed(fr[-3:])))] += 1
        if name.startswith(SKIP):
            return out
        o = out[0] if isinstance(out, (tuple, list)) and out else out
        if not (isinstance(o, torch.Tensor) and o.is_cuda):
            return out
        frames = [f for f in traceback.extract_stack() if "dtt_amd/" in f.filename or f.filename.endswith("bench.py")]
        where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in reversed(frames[-3:]))
        k = (name, where or "(autograd engine)")
        self.agg[k][0] += 1
        self.agg[k][1] += o.numel() * o.element_size()
        return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "xattn60m"
    args = argparse.Namespace(batch=None, precision="bf16", graph="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.autograd.set_multithreading_enabled(False)
    from dtt_amd.dist import GradBucketReducer
    w = bench.build_workload(name, args, dev, 0, 1, GradBucketReducer)
    for i in range(3):
        w["step"](i)
    torch.cuda.synchronize()
    nsteps = 2
    with Tally() as t:
        for i in range(nsteps):
            w["step"](3 + i)
        torch.cuda.synchronize()
    print("ops with a CPU tensor operand and a GPU result (a HIP graph would re-read that host memory at every replay):")
    for (op, where), c in t.h2d.most_common():
        print(f"    {c / nsteps:5.1f}/step  {op}  <-  {where}")
    if not t.h2d:
        print("    none")
    rows = sorted(t.agg.items(), key=lambda kv: -kv[1][1])
    print(f"workload {name}: {sum(v[0] for _, v in rows) / nsteps:.0f} ATen calls/step writing "
          f"{sum(v[1] for _, v in rows) / nsteps / 1e6:.1f} MB/step")
    print(f"{'MB/step':>9} {'n/step':>7}  op  <-  innermost dtt_amd frames")
    for (op, where), (n, by) in rows[:90]:
        print(f"{by / nsteps / 1e6:9.2f} {n / nsteps:7.1f}  {op}  <-  {where}")


if __name__ == "__main__":
    main()
