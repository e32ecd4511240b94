"""The procedural sky (SURVEY.md 8f N2 leftovers; Rtxpt/Lighting/Distant/SampleProceduralSky.{h,cpp,hlsli}, precomputed_sky.hlsli, EnvMapBaker.hlsl:228-236, 247-265):
  * pt_procedural_sky_update against an independent numpy restatement of SampleProceduralSky::Update: presets, the free-running clock, the low-pass filtered presets,
    the "changed" flag, the solid angle in double precision;
  * the oracle's restatement of the shader side (oracle/ptref/sky.h) against the reference's own text, live where /root/reference exists: whole cubes, bit for bit —
    sky alone, sky over an image with baked discs and the BC6H round trip, the sun inside the frame, the sun below the horizon (tests/test_env_cube.py holds the
    committed reference-text cubes of two cases for every machine);
  * properties: a pitch-black sky bakes to zero, switching the sky off restores the image-only cube, the texel functions are finite and non-negative."""
import math, os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rtxpt_amd as pt
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes

HAVE_REF = os.path.isdir("/root/reference/Rtxpt/Shaders")
PI_f = np.float32(3.141592654)


def _expected_constants(time, tod, p):
    f = np.float32
    c = {}
    c["FinalRadianceMultiplier"] = [f(f(f(p.brightness) * f(t)) * f(p.sunBrightness)) for t in p.colorTint]
    c["CloudsTime"] = f(math.fmod(time * float(f(p.cloudsMovementSpeed)), 86400.0))
    c["SunAngularDiameter"] = f(f(f(p.sunAngularDiameterDeg) / f(180.0)) * PI_f)
    c["sun_solid_angle"] = f(f(2) * PI_f) * f(1.0 - math.cos(0.5 * float(c["SunAngularDiameter"])))
    a = f(tod) * PI_f
    v = np.array([np.cos(a, dtype=f), f(0), np.sin(a, dtype=f)], np.float64); v /= np.linalg.norm(v)
    def rx(t): return np.array([[1, 0, 0], [0, math.cos(t), math.sin(t)], [0, -math.sin(t), math.cos(t)]])
    def ry(t): return np.array([[math.cos(t), 0, -math.sin(t)], [0, 1, 0], [math.sin(t), 0, math.cos(t)]])
    def rz(t): return np.array([[math.cos(t), math.sin(t), 0], [-math.sin(t), math.cos(t), 0], [0, 0, 1]])
    c["SunDir"] = v @ rx(-0.8) @ ry(-1.1) @ rz(math.radians(float(p.sunEastWestRotation)))      # Donut: row vectors, x then y then z
    return c


def test_update_matches_an_independent_restatement():
    p = pt.procedural_sky_default_params()
    assert np.allclose(list(p.colorTint), (1.45, 1.29, 1.27)) and p.sunBrightness == 5.0 and abs(p.sunAngularDiameterDeg - 0.5332) < 1e-7 and p.cloudDensityOffset == 0.75
    targets = {"MORNING": -0.25, "MIDDAY": 0.1, "EVENING": 0.51, "DAWN": 0.63, "PITCHBLACK": 1.0}
    for name, tod in targets.items():
        st = pt.PtProceduralSkyState()
        c, changed = pt.procedural_sky_update(st, 12.5, "==PROCEDURAL_SKY_%s==" % name, force_instant=True)
        assert changed
        e = _expected_constants(12.5, tod, p)
        want = [0, 0, 0] if name == "PITCHBLACK" else e["FinalRadianceMultiplier"]
        assert np.allclose(list(c.FinalRadianceMultiplier), want, rtol=1e-6)
        assert np.allclose(list(c.SunDir), e["SunDir"], atol=3e-7) and abs(np.linalg.norm(list(c.SunDir)) - 1) < 1e-6
        assert abs(c.CloudsTime - e["CloudsTime"]) < 1e-5 and c.SunAngularDiameter == e["SunAngularDiameter"] and abs(c.sun_solid_angle / e["sun_solid_angle"] - 1) < 1e-6
        assert c.PlanetSurfaceRadius == 6360.0 and c.PlanetAtmosphereRadius == 6420.0 and c.SqDistanceToHorizontalBoundary == 766800.0 and c.AtmosphereHeight == 60.0
        assert np.allclose(list(c.StarIrradiance), np.float32([1.47399998, 1.85039997, 1.91198003]) * np.float32(5.0)) and c.StarAngularDiameter == c.SunAngularDiameter
        assert list(c.GroundAlbedo) == [np.float32(0.3)] * 3 and list(c.physical_sky_ground_radiance) == [np.float32(0.00655480893)] * 3      # the two comma expressions
        assert (c.sky_transmittance, c.sky_scattering, c.cloud_density_offset) == (2.5, 2.0, 0.75) and abs(c.sky_phase_g - 0.9) < 1e-7 and abs(c.sky_amb_phase_g - 0.3) < 1e-7
        c2, changed2 = pt.procedural_sky_update(st, 12.5, "==PROCEDURAL_SKY_%s==" % name, force_instant=True)
        assert not changed2 and bytes(c2) == bytes(c)
    # the free-running clock: time of day = fmod(t * speed / 86400 + offset + 1, 2) - 1
    st = pt.PtProceduralSkyState()
    for t in (0.0, 100.0, 4000.0, 86400.0 * 3 + 17.0):
        c, _ = pt.procedural_sky_update(st, t)
        tod = np.float32(math.fmod((t * float(np.float32(300.0))) / float(np.float32(86400)) + float(np.float32(-0.4)) + 1.0, 2.0)) - np.float32(1.0)
        assert np.allclose(list(c.SunDir), _expected_constants(t, tod, p)["SunDir"], atol=3e-6), t
        assert st.timeOfDayL1 == st.timeOfDayL2 == tod
    # a preset without the instant update approaches its target through two low-pass filters: k = 1 - exp(-|clamp(t, 0, 0.3) * 0.1|) per call (m_lastSceneTime is never written)
    st = pt.PtProceduralSkyState(); l1 = l2 = np.float32(0)
    for i in range(5):
        c, _ = pt.procedural_sky_update(st, 10.0 + i, "==PROCEDURAL_SKY_EVENING==")
        k = np.float32(1.0) - np.float32(math.exp(-abs(float(np.float32(0.3) * np.float32(0.1)))))
        l1 = l1 + (np.float32(0.51) - l1) * k; l2 = l2 + (l1 - l2) * k
        assert abs(st.timeOfDayL1 - l1) < 1e-6 and abs(st.timeOfDayL2 - l2) < 1e-6
        assert np.allclose(list(c.SunDir), _expected_constants(0, l2, p)["SunDir"], atol=3e-6)
    with pytest.raises(RuntimeError): pt.procedural_sky_update(pt.PtProceduralSkyState(), 0.0, "some_image.exr")
    with pytest.raises(RuntimeError): pt.procedural_sky_update(pt.PtProceduralSkyState(), 0.0, "==PROCEDURAL_SKY_NOON==")


def _scene(preset, time, dim, image=False, **kw):
    consts, _ = pt.procedural_sky_update(pt.PtProceduralSkyState(), time, preset, force_instant=True)
    sc, cam = scenes.cornell_box("C2"); sc = dict(sc)
    if not image: sc["env"] = None
    sc["sky"] = {"consts": consts, "textures": pin_scenes._sky_textures()}; sc["env_cube_dim"] = dim; sc.update(kw)
    return sc, consts


@pytest.mark.skipif(not HAVE_REF, reason="no /root/reference on this machine: the reference text cannot be compiled here")
@pytest.mark.parametrize("preset,time,dim,image,kw", [("==PROCEDURAL_SKY_MIDDAY==", 0.0, 256, False, {}), ("==PROCEDURAL_SKY_EVENING==", 7.0, 64, False, {}),
                                                      ("==PROCEDURAL_SKY_DAWN==", 0.0, 64, True, {"env_compression": 1}), ("==PROCEDURAL_SKY==", 70000.0, 32, False, {}),
                                                      ("==PROCEDURAL_SKY_MORNING==", 3.0, 128, True, {})])
def test_oracle_sky_cube_matches_live_reference_text(preset, time, dim, image, kw):
    sc, consts = _scene(preset, time, dim, image, **kw)
    o = ptref.Oracle(reference_integrator=True, settings=scenes.default_settings()); o.set_scene(sc)
    a, d, lv = o.env_cube(); b, _, _ = o.env_cube(reference=True)
    bad = (a != b).any(-1)
    assert not bad.any(), "%d of %d texels differ from the reference-text bake" % (int(bad.sum()), bad.size)
    h = a.view(np.float16).astype(np.float32).reshape(-1, 4)
    assert np.isfinite(h).all() and (h >= 0).all() and h[:, :3].max() > 0
    if dim == 256: assert h[:, :3].max() > 1000.0, "the sun disc is inside a 256 cube's texel grid at midday"
    o.close()


def test_sky_properties():
    sc, consts = _scene("==PROCEDURAL_SKY_PITCHBLACK==", 0.0, 16)
    o = ptref.Oracle(); o.set_scene(sc)
    assert not o.env_cube()[0].view(np.float16).astype(np.float32).reshape(-1, 4)[:, :3].any(), "FinalRadianceMultiplier = 0 bakes to black"
    # switching the sky off restores the image-only cube
    sc2, _ = _scene("==PROCEDURAL_SKY_MIDDAY==", 0.0, 32, image=True)
    plain = dict(sc2); del plain["sky"]
    o2 = ptref.Oracle(); o2.set_scene(plain); want = o2.env_cube()[0].copy()
    o3 = ptref.Oracle(); o3.set_scene(sc2); withsky = o3.env_cube()[0].copy()
    assert (withsky != want).any()
    o3.set_procedural_sky(None); assert np.array_equal(o3.env_cube()[0], want)
    # the texel functions on random upper-hemisphere directions: finite, non-negative, transmittance within [0, 1]
    rng = np.random.default_rng(4); d = rng.normal(size=(4000, 3)); d[:, 2] = np.abs(d[:, 2]); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rows = np.concatenate([rng.integers(0, 64, (4000, 2)), rng.integers(0, 6, (4000, 1)), d], 1).astype(np.float32)
    low = o3.sky_eval(0, rows) if False else None
    o4 = ptref.Oracle(); o4.set_scene(sc2)
    low = o4.sky_eval(0, rows); atm = o4.sky_eval(1, rows)
    assert np.isfinite(low).all() and (low >= 0).all() and (low[:, 3] <= 1.0).all() and low[:, 3].min() < 0.999, "some directions go through clouds"
    assert np.isfinite(atm).all() and (atm >= 0).all() and atm.max() > 0
    pts = (np.float32([0, 0, 6360.1]) + d * rng.uniform(5, 60, (4000, 1))).astype(np.float32)
    tp = o4.sky_eval(2, np.concatenate([rows[:, :3], pts], 1))
    assert np.isfinite(tp).all() and (tp[:, 3:] >= 0).all() and (tp[:, 3:] <= 1.0).all()
    for x in (o, o2, o3, o4): x.close()
