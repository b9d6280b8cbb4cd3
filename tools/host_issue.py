#!/usr/bin/env python3
"""Host-only cost of issuing one training step, eager launches against one HIP-graph replay.

The number bench.py used to report as ``host_enqueue_ms_per_step`` (time until the last launch call of K
back-to-back steps returns) is NOT the host's cost once the GPU is the bottleneck: the launch calls block on the
full hardware queue and the figure converges to the GPU time.  Here each step is issued on an IDLE stream
(synchronise, issue, stamp, synchronise), so the stamp is what the host alone needs.

    python tools/host_issue.py [--workload xattn60m] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
import bench  # noqa: E402


def measure(mode, workload, reps):
    args = argparse.Namespace(batch=None, precision="bf16", graph=mode)
    from dtt_amd.dist import GradBucketReducer
    dev = torch.device("cuda", 0)
    w = bench.build_workload(workload, args, dev, 0, 1, GradBucketReducer)
    for i in range(3):
        w["step"](i)
    torch.cuda.synchronize()
    issue, total = [], []
    for i in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w["step"](i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    issue.sort(), total.sort()
    return {"launch": w["mode"], "host_issue_ms": round(issue[len(issue) // 2], 2),
            "step_ms_from_idle": round(total[len(total) // 2], 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="xattn60m")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    out = {"workload": a.workload}
    for mode in ("0", "1"):
        out["eager" if mode == "0" else "graph"] = measure(mode, a.workload, a.reps)
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
