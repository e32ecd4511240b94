"""The realtime mode's coupled frame (tests/realtime_cases.py): the ORACLE against the REFERENCE'S TEXT on the CPU.
  * committed runs (tests/golden/realtime_golden.npz, made by tests/golden/make_realtime_golden.py from LightsBaker.hlsl thread by thread and PathTracer.hlsli & co. in both
    PATH_TRACER_MODEs): every frame's tile tables, jitter, global proxy counters, feedback reservoirs after the fill passes, the planes' noisy radiance, specular hit distances and
    the build pass's depth / motion vectors / header — bit for bit;
  * one case against the live text where /root/reference exists (the generator's own code path);
  * the motion vectors matter: with the camera moving, the reprojected run differs from a run that reads history at the same pixel."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import realtime_cases as rc
import make_realtime_golden as gen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "realtime_golden.npz")


def _same(a, b): return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


@pytest.mark.parametrize("name", list(rc.cases()))
def test_oracle_matches_the_committed_reference_text_run(name):
    g = np.load(GOLDEN)
    got = gen.run_oracle(name, False)
    keys = list(got)
    assert keys and all(k in g.files for k in keys)
    bad = [k for k in keys if not _same(got[k], g[k])]
    assert not bad, bad[:8]
    frames = rc.cases()[name][4]
    filled = [int((g["%s_fbw%d" % (name, f)] > 0).sum()) for f in range(frames)]
    assert filled[-1] > filled[0] > 0, "the fill passes feed the reservoirs and the history carries over"
    assert all(((g["%s_motion_vectors%d" % (name, f)][..., :2] & 0x7FFF) != 0).any() for f in range(1, frames)), "the camera moves: the motion vectors are not zero"


@pytest.mark.skipif(not os.path.isdir("/root/reference/Rtxpt/Shaders"), reason="needs /root/reference (the build container)")
def test_live_reference_text_matches_the_fixture():
    name = "zoo_realtime"
    g = np.load(GOLDEN); got = gen.run_oracle(name, True)
    bad = [k for k in got if not _same(got[k], g[k])]
    assert not bad, bad[:8]


def test_the_motion_vectors_are_used():
    """The same run with the build pass's motion vectors zeroed before UpdateEnd (history read at the same pixel): the tile tables of later frames differ."""
    from oracle import ptref
    from rtxpt_amd import scenes
    name = "bistro_like_realtime"
    make, _, w, h, frames, subs, step, kw = rc.cases()[name]; S = rc.settings_for(name)
    sc, cam = make()
    o = ptref.Oracle(lp16=bool(int(S["useFp16Types"]))); o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.set_neeat(True)
    def read(frame):
        t, j, pc = o.neeat_tables(); return dict(table=t)
    out = rc.run(name, o.neeat_update_begin, lambda s, prm: o.build_stable_planes(s, prm), lambda fr: o.neeat_update_end(fr["depth"], np.zeros_like(fr["motion_vectors"])),
                 lambda s, prm, fr: o.fill_stable_planes(s, prm, fr), read, o.set_camera)
    g = np.load(GOLDEN)
    assert _same(out["%s_table0" % name], g["%s_table0" % name])      # frame 0: no history, nothing to reproject
    for f in range(1, frames): assert not _same(out["%s_table%d" % (name, f)], g["%s_table%d" % (name, f)]), f      # from frame 1 on the reservoirs are fetched from where the pixels were
    o.close()


def test_object_motion_is_in_the_animated_run():
    """The animated case without a previous pose in the scene (object motion reads as zero, the camera's stays): the build pass's motion vectors differ from the committed run from
    frame 0 on — the fixture's motion vectors do carry the objects' motion — while depth and header, which do not ask for the previous pose, stay."""
    from oracle import ptref
    from rtxpt_amd import scenes
    name = "bistro_like_c5_realtime_animated"
    make, _, w, h, frames, subs, step, kw = rc.cases()[name]; S = rc.settings_for(name)
    sc, cam = make()
    o = ptref.Oracle(lp16=bool(int(S["useFp16Types"]))); o.set_scene(sc); o.set_camera(scenes.bridge_camera(w, h, **cam)); o.set_settings(S); o.resize(w, h); o.set_neeat(True)
    def pose(cur, prev):
        posed = dict(sc); posed["instances"], posed["positions"] = cur; o.set_scene(posed)
    def read(frame): return dict(motion_vectors=frame["motion_vectors"], depth=frame["depth"], header=frame["header"])
    out = rc.run(name, o.neeat_update_begin, lambda s, prm: o.build_stable_planes(s, prm), lambda fr: o.neeat_update_end(fr["depth"], fr["motion_vectors"]),
                 lambda s, prm, fr: o.fill_stable_planes(s, prm, fr), read, o.set_camera, pose)
    g = np.load(GOLDEN)
    for f in range(frames):
        a, b = out["%s_motion_vectors%d" % (name, f)], g["%s_motion_vectors%d" % (name, f)]
        moved = (np.asarray(a) != np.asarray(b)).reshape(h, w, -1).any(-1)
        assert 20 < moved.sum() < moved.size // 2, (f, int(moved.sum()))      # the animated props and the banner, not the street
        assert _same(out["%s_depth%d" % (name, f)], g["%s_depth%d" % (name, f)]) and _same(out["%s_header%d" % (name, f)], g["%s_header%d" % (name, f)])
    o.close()
