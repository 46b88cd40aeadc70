import importlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub: str = ""):
    """The product package directory is `u-llava_amd` (hyphen) -> importlib only."""
    return importlib.import_module("u-llava_amd" + ("." + sub if sub else ""))


def load_fixture(name: str):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)


def fixture_state_dict(fx, dtype=None):
    W = pkg("weights")
    dt = dtype or getattr(torch, fx["dtype"].split(".")[-1])
    sd32 = W.seeded_state_dict(fx["shapes"], fx["seed"], torch.float32)
    return {k: v.to(dt) for k, v in sd32.items()}


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()
