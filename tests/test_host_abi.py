"""CPU-side checks of the product library: it loads, exports every symbol include/mi355pt.h declares, refuses to run
without a GPU (no CPU fallback), and its host-only helpers match the reference-derived golden vectors."""
import ctypes
import os
import re

import numpy as np
import pytest

import rtxpt_amd as pt
from rtxpt_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(pt.LIB_PATH):
        pt.build_library()
    return pt.load_library()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mi355pt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", src)))


def test_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(pt.EXPORTS) == names


def test_shipped_library_carries_no_test_hooks(lib):
    """include/mi355pt_testhooks.h: the evaluation hooks live in libmi355pt_testhooks.so (same sources, -DMI355PT_TEST_HOOKS), the shipped library exports none of them."""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mi355pt_testhooks.h")).read(), flags=re.S)
    hooks = sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", src)))
    assert hooks == sorted(pt.TEST_HOOK_EXPORTS) and hooks
    for n in hooks:
        assert not hasattr(lib, n), "the shipped library exports the test hook " + n
    hl = pt.load_library(test_hooks=True)
    for n in declared_symbols() + hooks:
        assert hasattr(hl, n), "libmi355pt_testhooks.so misses " + n


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError) as e:
        pt.PathTracer()
    assert e.value.code == 2          # PT_ERROR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rtxpt_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in txt.replace("CPU oracle", "") or f == "__init__.py" and "oracle/" not in txt, (f, "product source mentions oracle/")
                assert "import oracle" not in txt and "from oracle" not in txt and "ptref" not in txt, f


def test_bridge_camera_matches_reference_golden(lib, golden):
    for c in golden["bridge_camera"]:
        cam = pt.bridge_camera(c["w"], c["h"], c["pos"], c["dir"], c["up"], c["fov"], c["near"], c["far"], c["focal"], c["aperture"], c["jitter"])
        ref = np.frombuffer(bytes.fromhex(c["bytes"]), dtype=scenes.CAMERA_DTYPE)[0]
        for name in scenes.CAMERA_DTYPE.names:
            if name.startswith("_"):
                continue
            a, b = np.asarray(cam[name]), np.asarray(ref[name])
            assert np.array_equal(a, b), (name, a, b)       # same libm (std::tan/atan), same operation order: bit exact


def test_default_settings(lib):
    s = np.zeros((), dtype=scenes.SETTINGS_DTYPE)
    assert lib.pt_default_settings(s.ctypes.data_as(ctypes.c_void_p)) == 0
    d = scenes.default_settings(useFp16Types=1)          # the C entry point returns the reference's default build of the lp types (SampleUI.h:182)
    for n in scenes.SETTINGS_DTYPE.names:
        assert np.array_equal(s[n], d[n]), n
    assert lib.pt_default_settings(None) == 1
