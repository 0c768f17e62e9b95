import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun); everything else runs on CPU")


def load_golden(case):
    with open(os.path.join(GOLDEN, case + ".json")) as f:
        meta = json.load(f)
    arrays = dict(np.load(os.path.join(GOLDEN, case + ".npz")))
    return meta, arrays


def golden_problem(case):
    """(meta, golden arrays, topology, numpy state dict, feat, cand) of a golden case --
    weights and inputs are regenerated from the seeds, only outputs are stored."""
    from livespeechportraits_amd import synth
    from livespeechportraits_amd.topology import build_topology
    meta, arrays = load_golden(case)
    topo = build_topology(meta["variant"], ngf=meta["ngf"], num_downs=meta["num_downs"], size=meta["size"])
    sd = synth.make_state_dict(topo, meta["weight_seed"])
    feat, cand = synth.make_inputs(meta["batch"], meta["size"], meta["input_seed"], meta["cand_batch"])
    return meta, arrays, topo, sd, feat, cand


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a ROCm device")
    return torch.device("cuda:0")
