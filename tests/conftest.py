import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-separation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def load_manifest(name):
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return {k: tuple(v) for k, v in json.load(f)[name].items()}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def manifest():
    return load_manifest


def rel_err(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _pack_plan_off():
    """dtt_amd.GraphedTrainStep and bench.build_workload switch the process-wide pack plan (ops._PackPlan) on, tests
    switch the side streams: never into the next test."""
    yield
    mod = sys.modules.get("dtt_amd.ops")
    if mod is not None:
        mod.weight_plan(False)
        mod.handoff_kernels(True)
        mod.wgrad_lane(os.environ.get("DTT_WGRAD_LANE", "1") == "1")
        mod.video_lane(os.environ.get("DTT_VIDEO_LANE", "1") == "1")
