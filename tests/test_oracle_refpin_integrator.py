"""Oracle pinning, third layer: the oracle's INTEGRATOR against the reference's integrator source text.

oracle/refpin/hlsl_tu.py --integrator compiles PathTracer.hlsli and its whole include closure (HandleHit, HandleMiss, GenerateScatterRay, nested
dielectrics, NEE with its reservoir and MIS, Russian roulette, the packed PathState, LightSampler, PolymorphicLight, EnvMap, BxDF ...) from
/root/reference; oracle/refpin/hlsl_pt_wrappers.inc serves its `Bridge` from the oracle's scene services and runs the raygen loop. The frame that
comes out must equal the oracle's own frame bit for bit, with the same ray counts.

  * test_oracle_matches_reference_integrator_golden: committed frames (tests/golden/reference_integrator_golden.npz) — runs everywhere.
  * test_oracle_matches_live_reference_integrator: the libraries built here, where /root/reference exists (one per shader-macro combination)."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_integrator_golden.npz")
CASES = pin_scenes.cases()


def _oracle_frame(name, reference=False):
    make, S, w, h, first, n = CASES[name]
    sc, cam = make()
    o = ptref.Oracle(reference_integrator=True, settings=S) if reference else ptref.Oracle()
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.render(first, n)
    c = o.counters()
    return o.radiance(), (c["extendRays"], c["shadowRays"])


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_integrator_golden(name):
    g = np.load(GOLDEN)
    got, rays = _oracle_frame(name)
    want = g[name]
    assert got.shape == want.shape
    bad = (got.view(np.uint32) != want.view(np.uint32)).any(-1)
    assert not bad.any(), "%s: %d of %d pixels differ from the reference-text frame" % (name, int(bad.sum()), bad.size)
    assert tuple(int(v) for v in g[name + "_rays"]) == rays
    assert want[..., :3].max() > 0


@pytest.mark.parametrize("name", ["c1", "c2", "c2_nested2_norr_nold", "bistro_like_c5"])
def test_oracle_matches_live_reference_integrator(name):
    if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
        pytest.skip("no /root/reference on this machine: the reference-text integrator cannot be built here")
    want, rays_ref = _oracle_frame(name, reference=True)
    got, rays = _oracle_frame(name)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and rays == rays_ref
