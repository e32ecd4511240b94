"""pt_set_light_importance_boost on the device (run with -m gpu): the frustum term of LightsBaker's ImportanceBooster in k_light_weights. The oracle's restatement is pinned to the
reference text on the CPU (tests/test_light_importance_boost.py); here the device's proxy table, frames and NEE-AT runs with the boost equal the oracle's."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rtxpt_amd import scenes

pytestmark = pytest.mark.gpu


def _pair(S, w, h, scale=0.02):
    import rtxpt_amd as pt
    from oracle import ptref
    sc, cam = scenes.bistro_like(scale=scale, tex_size=64)
    camd = scenes.bridge_camera(w, h, **cam)
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(camd); t.resize(w, h)
    o = ptref.Oracle(lp16=bool(S["useFp16Types"])); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h)
    return t, o, cam


def test_proxy_table_and_frame_with_the_boost():
    S = scenes.default_settings(); w, h = 96, 54
    t, o, cam = _pair(S, w, h)
    plain = t.lights()["proxyCounters"].copy()
    M = scenes.view_projection(w, h, **cam)
    t.set_light_importance_boost(M); o.set_light_importance_boost(M)
    o.L.ptref_prepare(o.h)
    a, b = t.lights(), o.lights()
    assert np.array_equal(a["proxyCounters"], b["proxyCounters"]) and np.array_equal(a["proxyIndices"], b["proxyIndices"]) and not np.array_equal(a["proxyCounters"], plain)
    t.render(0, 2); o.render(0, 2)
    assert np.array_equal(t.radiance().view(np.uint32), o.radiance().view(np.uint32))
    # another camera: only weights and proxies follow (the lights are not re-baked), still equal to the oracle's full bake
    cam2 = dict(cam, pos=tuple(np.asarray(cam["pos"]) + np.array([3.0, 0.5, -2.0])))
    M2 = scenes.view_projection(w, h, **cam2); camd2 = scenes.bridge_camera(w, h, **cam2)
    bake0 = t.build_stats()["lightBakeMs"]
    t.set_camera(camd2); t.set_light_importance_boost(M2); o.set_camera(camd2); o.set_light_importance_boost(M2); o.L.ptref_prepare(o.h)
    a2 = t.lights()
    assert t.build_stats()["lightBakeMs"] == bake0 and np.array_equal(a2["proxyCounters"], o.lights()["proxyCounters"]) and not np.array_equal(a2["proxyCounters"], a["proxyCounters"])
    t.set_light_importance_boost(None); assert np.array_equal(t.lights()["proxyCounters"], plain)      # off again
    with pytest.raises(Exception, match="negative"): t.set_light_importance_boost(M, mul=-1.0)
    t.close()


def test_neeat_run_with_the_boost():
    """the reference's default light sampling in full: frustum boost + usage feedback + tile tables, three frames, device == oracle on everything"""
    S = scenes.default_settings(NEEType=2, useFp16Types=1); w, h = 128, 72
    t, o, cam = _pair(S, w, h)
    M = scenes.view_projection(w, h, **cam)
    t.set_light_importance_boost(M); o.set_light_importance_boost(M); t.set_neeat(True); o.set_neeat(True)
    for f in range(3):
        t.render(f, 1); o.render(f, 1)
        (td, jd), (to, jo, pco) = t.neeat_tables(), o.neeat_tables()
        assert jd == jo and np.array_equal(td, to) and np.array_equal(pco, t.lights()["proxyCounters"]), "frame %d" % f
        (wd, cd), (wo, co) = t.light_feedback(0), o.light_feedback(0)
        assert np.array_equal(wd.view(np.uint32), wo.view(np.uint32)) and np.array_equal(cd, co), "frame %d: reservoirs" % f
    assert np.array_equal(t.radiance().view(np.uint32), o.radiance().view(np.uint32))
    t.close()
