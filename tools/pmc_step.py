#!/usr/bin/env python3
"""HBM traffic of the step's largest entry points, counted INSIDE an eagerly issued training step: the calls of the
top entries (by time, dtt_amd.prof.CallTimer) are bracketed by marker launches (a tiny fp32 dtt_binary whose grid
size encodes the entry), so that a rocprofv3 counter pass can attribute every dispatch between two markers:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc3_rd -- python3 tools/pmc_step.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc3_wr -- python3 tools/pmc_step.py
    python3 tools/pmc_step_parse.py gpurun_out/pmc3_rd gpurun_out/pmc3_wr      -> profiles/r03_pmc_traffic.{json,txt}

(separate passes: the two counters do not fit one TCC pass).  Run it once WITHOUT the profiler first: that run ranks
the entries by their un-profiled time and writes the marker table (gpurun_out/pmc_step_keys.json), which the two
counter passes reuse so that both mark the same entries (tools/r03_profile.sh does all of it)."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd"))
import bench  # noqa: E402
from dtt_amd import _lib, prof  # noqa: E402
from dtt_amd.dist import GradBucketReducer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="xattn60m")
ap.add_argument("--top", type=int, default=12)
ap.add_argument("--keys", default=os.path.join(ROOT, "gpurun_out", "pmc_step_keys.json"))
a = ap.parse_args()
args = argparse.Namespace(batch=None, precision="bf16", graph="0")
from dtt_amd import ops  # noqa: E402
ops.wgrad_lane(False)
ops.video_lane(False)
w = bench.build_workload(a.workload, args, torch.device("cuda", 0), 0, 1, GradBucketReducer)
for i in range(2):
    w["step"](i)
torch.cuda.synchronize()


def row_key(entry, dims):
    return bench._traffic_key(dict(entry=entry, shape=dict(dims)))


if os.path.exists(a.keys):
    table = json.load(open(a.keys))
else:
    with prof.CallTimer() as ct:
        w["step"](2)
    rows = [r for r in ct.table(1)["rows"] if "frac" in r][:a.top]
    table = {row_key(r["entry"], r["shape"]): dict(marker=i + 1, entry=r["entry"], shape=r["shape"],
                                                   launches_per_step=r["launches_per_step"],
                                                   algorithmic_bytes=r.get("bytes_per_launch"), flops=r.get("flops_per_launch"))
             for i, r in enumerate(rows)}
    os.makedirs(os.path.dirname(a.keys), exist_ok=True)
    json.dump(table, open(a.keys, "w"), indent=1)

mbuf = torch.zeros(3, 64 * 1024, device="cuda")
real = _lib.lib()


def mark(k):                                             # grid = k + 1 workgroups of binary_kernel<float, 0>
    real.dtt_binary(0, C.c_void_p(mbuf[0].data_ptr()), C.c_void_p(mbuf[1].data_ptr()), C.c_void_p(mbuf[2].data_ptr()),
                    C.c_int64((k + 1) * 1024), 0, _lib.stream_ptr())


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        spec = prof.SPECS.get(name)
        if spec is None:
            return fn

        def call(*cargs):
            try:
                dims, _, _ = spec(cargs)
                hit = table.get(row_key(name, dims))
            except Exception:                                 # noqa: BLE001
                hit = None
            if hit is None:
                return fn(*cargs)
            mark(hit["marker"])
            rc = fn(*cargs)
            mark(0)
            return rc
        return call


_lib._lib = Proxy()
w["step"](3)
torch.cuda.synchronize()
_lib._lib = real
print("marked", len(table), "entries")
