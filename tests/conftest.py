import os
import sys

import pytest

# The suite drives the wavefront kernels on frames of a few thousand paths, which the product would hand to the tail kernel (pt_set_tail_paths, default 65536) from the
# first pass on. So contexts made by the tests start with the tail kernel OFF unless a test asks for it: tests/test_gpu_tail_kernel.py renders the same pinned frames
# with it ON (whole frames, mixed frames, the hand-back path); smoke() and bench.py's parity block run the product default.
os.environ.setdefault("MI355PT_TAIL_PATHS", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ptref
    return ptref.lib()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "refpin_golden.json")) as f:
        return json.load(f)
