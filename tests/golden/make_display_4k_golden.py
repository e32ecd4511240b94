"""Generates tests/golden/display_4k_golden.npz: the display path at full size — the bench frame as the REFERENCE'S integrator text renders it (make_bench_frame_golden.py's workload),
through applyToneMapping for the six operators at three exposure compensations and with auto exposure: the reference's ToneMapping.ps.hlsli text and the oracle's restatement agree on
every one of the 8.3 M pixels (checked here, asserted), and the SRGBA8_UNORM store of the result is kept as SHA-256 per parameter set. tests/test_gpu_full_size.py compares the
device's pt_tonemap of its own frame with it. Run in the build container only (two minutes of reference text for the frame):   python tests/golden/make_display_4k_golden.py"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import rtxpt_amd as pt
from rtxpt_amd import scenes
from oracle import ptref
import make_bench_frame_golden as gen


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def parameter_sets():
    out = {"op%d_ec%+d" % (op, ec): dict(exposure_compensation=float(ec), toneMapOperator=op) for op in range(6) for ec in (-2, 0, 3)}
    out["auto_exposure"] = dict(autoExposure=1, avgLuminance=0.3)
    return out


if __name__ == "__main__":
    cache = "/tmp/bench_ref_frame.npy"
    if os.path.exists(cache): rad = np.load(cache)
    else:
        sc, cam, S = gen.bench_workload()
        o = ptref.Oracle(reference_integrator=True, settings=S, lp16=True); o.set_scene(sc); o.set_camera(scenes.bridge_camera(gen.W, gen.H, **cam)); o.set_settings(S); o.resize(gen.W, gen.H)
        o.render(0, gen.SPP); rad = o.radiance().copy()
    assert np.array_equal(digest(rad), np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_frame_golden.npz"))["sha256"])
    out = {}
    for name, kw in parameter_sets().items():
        t = pt.default_tonemap(**kw)
        a = ptref.tonemap_linear(rad, t, reference=True); b = ptref.tonemap_linear(rad, t, reference=False)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name            # the reference's text == the restatement, every pixel
        out[name] = digest(ptref.tonemap(rad, t)); print(name, "ok", flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "display_4k_golden.npz"), **out)
