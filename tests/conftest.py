import os
import sys

import pytest
import torch

# The CPU oracle runs thousands of small-tensor ops (a 1 s clip is 60 x 192 activations): on a 256-thread host torch's default — one thread per
# logical CPU — spends its time handing out work (the 1000-step chain of test_gpu_parity.py took 146-222 s by box).  16 threads are as fast as
# any count here and keep the suite's duration independent of the host.
torch.set_num_threads(min(16, torch.get_num_threads()))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(scope="session")
def unet_sd():
    from said_amd.util import synth
    return synth.fill_state_dict(synth.unet_param_shapes())


@pytest.fixture(scope="session")
def w2v_sd():
    from said_amd.util import synth
    return synth.fill_state_dict(synth.w2v_param_shapes())
