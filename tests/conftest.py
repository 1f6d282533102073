import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU suite is batch-1 oracle work (per-frame LSTM mat-vecs): half the cores is as fast as all of them on an idle
    # host and several times faster on a busy one (oversubscribed oneDNN threads spin on each other).
    import torch
    if not torch.cuda.is_available():
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 2) // 2)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synth_assets():
    """seeded weights + body shared by every test (regenerated, never stored)."""
    from robustcap_amd import synth
    return {"state_dict": synth.make_state_dict(0), "body": synth.make_body(1)}
