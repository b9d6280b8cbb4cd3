#!/usr/bin/env python3
"""cProfile of the HOST side of eagerly issued training steps (where the ~1500 launches per step are issued from).
usage: python tools/host_profile.py [workload] [steps]"""
import argparse
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
import bench  # noqa: E402
from dtt_amd.dist import GradBucketReducer  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "xattn60m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
args = argparse.Namespace(batch=None, precision="bf16", graph="0")
w = bench.build_workload(wl, args, torch.device("cuda", 0), 0, 1, GradBucketReducer)
for i in range(3):
    w["step"](i)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)      # backward on this thread, where cProfile can see it
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    w["step"](i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
