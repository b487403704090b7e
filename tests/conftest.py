import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gpu-acceleration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    import json
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return [c["name"] for c in json.load(f)["cases"]]


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, f"msm_{name}.npz"))
