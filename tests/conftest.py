import os
import sys

import pytest

# The suite drives the wavefront kernels on frames of a few thousand paths, which the product would hand to the tail kernel (pt_set_tail_paths, default 4096 live paths per
# batch since round 6) from the first pass on. So contexts made by the tests start with the tail kernel OFF unless a test asks for it — except in the modules of TAIL_BOTH_WAYS
# below (the golden-digest tests against the reference's text at HD and full size, the sharded frames, NEE-AT with its exported depth): every test of those runs twice, once with
# the tail kernel off and once in the product's default configuration. tests/test_gpu_tail_kernel.py renders the pinned frames entirely through the tail kernel (thresholds up to
# 300 000); smoke() and bench.py's parity block run the product default. Fused traversal launches (pt_set_fused_traversal) are on in both; the separate launches are compared
# with them in tests/test_gpu_fused_traversal.py.
os.environ.setdefault("MI355PT_TAIL_PATHS", "0")
TAIL_BOTH_WAYS = ("test_gpu_reference_goldens", "test_gpu_parity_hd", "test_gpu_full_size", "test_gpu_multigpu_c4", "test_gpu_multigpu_c5", "test_gpu_neeat", "test_gpu_neeat_baker")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "tail_once(configuration): in a module of TAIL_BOTH_WAYS, run this test in the named tail-kernel configuration only")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ptref
    return ptref.lib()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "refpin_golden.json")) as f:
        return json.load(f)


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in TAIL_BOTH_WAYS and "tail_configuration" in metafunc.fixturenames:
        once = metafunc.definition.get_closest_marker("tail_once")      # @pytest.mark.tail_once("tail_default"): a long test that runs in ONE configuration (the named one)
        metafunc.parametrize("tail_configuration", [once.args[0]] if once else ["tail_off", "tail_default"], indirect=True)


@pytest.fixture(autouse=True)
def tail_configuration(request, monkeypatch):
    """MI355PT_TAIL_PATHS for the contexts a test creates (read at pt_create): 0 everywhere, and the product default as a second run in the modules of TAIL_BOTH_WAYS."""
    mode = getattr(request, "param", "tail_off")
    if mode == "tail_default": monkeypatch.delenv("MI355PT_TAIL_PATHS", raising=False)
    else: monkeypatch.setenv("MI355PT_TAIL_PATHS", "0")
    return mode
