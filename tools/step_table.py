#!/usr/bin/env python3
"""Per-entry-point, per-shape table of one training step (dtt_amd.prof.CallTimer over eagerly issued steps of a
bench.py workload).  usage: python tools/step_table.py [workload] [--json out.json] [--steps 3]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
import bench  # noqa: E402
from dtt_amd import prof  # noqa: E402
from dtt_amd.dist import GradBucketReducer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="xattn60m")
ap.add_argument("--json", default=None)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=None)
a = ap.parse_args()
args = argparse.Namespace(batch=a.batch, precision="bf16", graph="0")
from dtt_amd import ops  # noqa: E402
ops.wgrad_lane(False)          # one stream: beside another kernel an entry's bracket would measure the pair
ops.video_lane(False)
w = bench.build_workload(a.workload, args, torch.device("cuda", 0), 0, 1, GradBucketReducer)
for i in range(3):
    w["step"](i)
torch.cuda.synchronize()
with prof.CallTimer() as ct:
    for i in range(a.steps):
        w["step"](i)
tab = ct.table(a.steps)
print(prof.format_table(tab, top=60))
if a.json:
    with open(a.json, "w") as f:
        json.dump(tab, f, indent=1)
