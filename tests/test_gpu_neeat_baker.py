"""NEE-AT with the light baker in the loop, on the device (run with -m gpu): pt_set_neeat.

Every frame = the feedback passes as kernels (k_neeat_*: PreFilter from a snapshot, P0 with atomic usage counts, the proxy table rebuilt with the feedback term, P1a, P1b,
tile fill, bitonic sort + run lengths in LDS, Clear), then the wavefront path tracer with the deferred feedback insert.
  * against the REFERENCE TEXT's runs (tests/golden/neeat_loop_golden.npz), every frame: tile tables, jitter, global proxy counters, feedback reservoirs, and the frame;
  * against the oracle; one pt_render call tracing all frames against frame-by-frame calls; reset / disable."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from rtxpt_amd import scenes
import pin_scenes
import make_neeat_loop_golden as loop
from test_neeat_baker import compare, golden

pytestmark = pytest.mark.gpu
CASES = pin_scenes.neeat_loop_cases()


@pytest.mark.parametrize("name", list(CASES))
def test_device_run_matches_reference_text(name):
    want = golden(name)
    got = loop.run_device(name)
    assert set(got) == set(want)
    compare(name, got, want)


def test_device_run_matches_oracle():
    name = "c2_sphere_lights_loop_lp16"
    compare(name, loop.run_device(name), loop.run_oracle(name, False))


@pytest.mark.parametrize("name", ["bistro_like_loop", "bistro_like_c5_loop_nofilter"])
def test_one_call_equals_frame_by_frame(name):
    want = golden(name)
    got = loop.run_device(name, one_call=True)
    compare(name, got, want, keys=list(got))


def test_reset_and_disable():
    import rtxpt_amd as pt
    make, S, w, h, frames, opts = CASES["bistro_like_loop"]
    sc, cam = make()
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h)
    t.render(0, 1); plain = t.radiance().copy(); t.reset_accumulation()
    with pytest.raises(Exception): t.neeat_tables()
    t.set_neeat(True, **opts); t.render(0, 2); first = t.radiance().copy(); t1 = t.neeat_tables()[0].copy(); t.reset_accumulation()
    t.neeat_reset(); t.render(0, 2)
    assert np.array_equal(first.view(np.uint32), t.radiance().view(np.uint32)) and np.array_equal(t1, t.neeat_tables()[0])
    t.reset_accumulation(); t.render(0, 2)
    assert not np.array_equal(t1, t.neeat_tables()[0])
    t.set_neeat(False); t.reset_accumulation(); t.render(0, 1)
    assert np.array_equal(plain.view(np.uint32), t.radiance().view(np.uint32))
    with pytest.raises(Exception, match="0.95"): t.set_neeat(True, ratio=1.0)
    S3 = S.copy(); S3["NEEFullSamples"] = 2; t.set_neeat(True, **opts); t.set_settings(S3)
    with pytest.raises(Exception, match="NEEFullSamples 1"): t.render(0, 1)
    t.close()


def test_full_hd_run_is_sane():
    """1920 x 1080, 3 frames: the kernels at a size where the tile grid (241 x 136) and the feedback planes are not toys; tables sorted, counts consistent, jitter in range"""
    import rtxpt_amd as pt
    sc, cam = scenes.bistro_like(scale=0.05, tex_size=256)
    S = scenes.default_settings(NEEType=2)
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(1920, 1080, **cam)); t.resize(1920, 1080); t.set_neeat(True)
    t.render(0, 3)
    tab, jit = t.neeat_tables(); fw, fc = t.light_feedback(0)
    n = len(t.lights()["lights"])
    assert tab.shape == (136, 241, 128) and max(jit) < 8
    lights, counts = tab >> 9, (tab & 0x1FF) + 1
    assert lights.max() < n and (np.diff(lights.astype(np.int64), axis=-1) >= 0).all()
    run_start = np.concatenate([np.ones(lights.shape[:2] + (1,), bool), lights[..., 1:] != lights[..., :-1]], -1)
    assert (counts[run_start].astype(np.int64).reshape(-1).sum() == 128 * 136 * 241)      # the runs of a tile add up to its 128 entries
    assert (fw > 0).mean() > 0.5 and ((fc != 0xFFFFFFFF) == (fw > 0)).all() and (fc[fc != 0xFFFFFFFF] & 0x7FFFFFFF).max() < n
    assert np.isfinite(t.radiance()).all() and t.radiance()[..., :3].mean() > 0
    t.close()


@pytest.mark.parametrize("w,h,frames", [(640, 360, 3), (1440, 810, 2)], ids=["640x360", "1440x810_pipelined_batches"])
def test_device_run_matches_oracle_at_size(w, h, frames):
    """The same run on the oracle and on the device at sizes where the device works with thousands of tiles, atomics under contention and (1440 x 810: 1.17 M paths) several
    sub-frame batches on their own streams writing one set of feedback reservoirs: tile tables, jitter, proxy counters, reservoirs and the accumulated frame, bit for bit."""
    import rtxpt_amd as pt
    from oracle import ptref
    sc, cam = scenes.bistro_like(scale=0.05, tex_size=128)
    S = scenes.default_settings(NEEType=2, useFp16Types=1)
    camd = scenes.bridge_camera(w, h, **cam)
    o = ptref.Oracle(lp16=True); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.set_neeat(True)
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(camd); t.resize(w, h); t.set_neeat(True)
    for f in range(frames):
        o.render(f, 1); t.render(f, 1)
        to, jo, pco = o.neeat_tables(); td, jd = t.neeat_tables()
        assert jo == jd and np.array_equal(to, td), "frame %d: tile tables" % f
        assert np.array_equal(pco, t.lights()["proxyCounters"]), "frame %d: global proxy counters" % f
        (wo, co), (wd, cd) = o.light_feedback(0), t.light_feedback(0)
        assert np.array_equal(wo.view(np.uint32), wd.view(np.uint32)) and np.array_equal(co, cd), "frame %d: feedback reservoirs" % f
    bad = (o.radiance().view(np.uint32) != t.radiance().view(np.uint32)).any(-1)
    assert not bad.any(), "%d of %d pixels differ" % (int(bad.sum()), bad.size)
    t.close()


@pytest.mark.parametrize("w,h", [(1, 1), (3, 2), (8, 8), (9, 17), (33, 5)], ids=["1x1", "3x2", "8x8", "9x17", "33x5"])
def test_tiny_frames_device_matches_oracle(w, h):
    from test_neeat_baker import _tiny_run
    a, b = _tiny_run("device", w, h), _tiny_run("oracle", w, h)
    for f, (x, y) in enumerate(zip(a, b)): compare("%dx%d frame %d" % (w, h, f), x, y)


def test_edge_states():
    """No lights at all (NEE cannot run: the frames equal the plain ones and nothing is read from tables that do not exist); the manual table setter is refused while the
    baker runs the loop; a different light set (here: the analytic lights removed) restarts the feedback history instead of reading stale indices."""
    import rtxpt_amd as pt
    sc, cam = scenes.cornell_box("C1"); sc = dict(sc); sc["env"] = None
    dark = dict(sc); m = sc["materials"].copy(); m["EmissiveColor"][:] = 0; dark["materials"] = m
    S = scenes.config_settings("C1").copy(); S["NEEType"] = 2
    t = pt.PathTracer(); t.set_scene(dark); t.set_settings(S); t.set_camera(scenes.bridge_camera(48, 48, **cam)); t.resize(48, 48)
    if len(t.lights()["lights"]) == 0:
        t.render(0, 2); plain = t.radiance().copy(); t.reset_accumulation()
        t.set_neeat(True); t.render(0, 2)
        assert np.array_equal(plain.view(np.uint32), t.radiance().view(np.uint32))
    t.close()
    make, S2, w, h, frames, opts = CASES["c2_sphere_lights_loop_lp16"]
    sc2, cam2 = make()
    t = pt.PathTracer(); t.set_scene(sc2); t.set_settings(S2); t.set_camera(scenes.bridge_camera(w, h, **cam2)); t.resize(w, h); t.set_neeat(True, **opts)
    t.render(0, 2)
    with pytest.raises(Exception, match="pt_set_neeat is on"): t.set_local_light_sampling(np.zeros((6, 9, 128), np.uint32))
    n0 = len(t.lights()["lights"])
    sc3 = dict(sc2); sc3["lights"] = (sc2["lights"][0][:1], sc2["lights"][1][:1]); t.set_scene(sc3)      # fewer analytic lights: another light set
    t.render(2, 2)
    n1 = len(t.lights()["lights"]); assert n1 < n0
    tab, _ = t.neeat_tables(); assert (tab >> 9).max() < n1 and np.isfinite(t.radiance()).all()
    t.close()


def test_animated_run_keeps_its_history():
    """NEE-AT across animated frames (C5's seam: pt_animate -> refit + emissive re-bake between frames): the light set keeps its order, so the reservoirs and tiles of the last
    frame stay meaningful — device (refit, device-side re-bake) == oracle (instances set, everything rebuilt) on every frame."""
    import rtxpt_amd as pt
    from oracle import ptref
    sc, cam = scenes.bistro_like(scale=0.01, tex_size=64, animated=True)
    S = scenes.default_settings(NEEType=2, nestedDielectricsQuality=2); w, h = 128, 72
    camd = scenes.bridge_camera(w, h, **cam)
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(camd); t.resize(w, h); t.set_neeat(True)
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(w, h); o.set_neeat(True)
    filled = []
    for f in range(4):
        inst = scenes.animate_instances(sc, 0.35 * f)
        t.animate(instances=inst, rebuild=False); o.set_instances(inst); o.reset_accumulation()      # (pt_animate restarts the accumulation, as any scene change does in reference mode)
        t.render(f, 1); o.render(f, 1)
        (td, jd), (to, jo, pco) = t.neeat_tables(), o.neeat_tables()
        assert jd == jo and np.array_equal(td, to) and np.array_equal(pco, t.lights()["proxyCounters"]), "frame %d" % f
        (wd, cd), (wo, co) = t.light_feedback(0), o.light_feedback(0)
        assert np.array_equal(wd.view(np.uint32), wo.view(np.uint32)) and np.array_equal(cd, co), "frame %d: reservoirs" % f
        filled.append(int((wd > 0).sum()))
    assert filled[-1] > filled[0]                      # the history carried over the animated frames (a restart would look like frame 0 again)
    assert np.array_equal(t.radiance().view(np.uint32), o.radiance().view(np.uint32))
    t.close()


@pytest.mark.parametrize("view_projection", [False, True], ids=["same-pixel-history", "view-projection"])
def test_tile_sharded_run_equals_the_unsharded_run(view_projection):
    """Three ranks of a tile-sharded frame on one device, the host moving the packed reservoirs and exported depth between frames (pt_neeat_pack_feedback -> pt_neeat_unpack_feedback;
    with a communicator pt_render does the same through RCCL): every rank ends up with the unsharded run's tile tables, proxy counters and reservoirs on every frame, and the
    gathered frame is the unsharded frame. With pt_set_view_projection the baker's reprojection compares the exported depth of this frame and the last — also of pixels another
    rank traced (neighbourhoods cross shard borders): the camera moves between the frames so that pixels do get disoccluded."""
    import torch
    import rtxpt_amd as pt
    sc, cam = scenes.bistro_like(scale=0.02, tex_size=128)
    S = scenes.default_settings(NEEType=2, useFp16Types=1); w, h, frames, world = 200, 120, 4, 3
    def camera(f):
        c = dict(cam)
        if view_projection: c["pos"] = tuple(np.asarray(cam["pos"], np.float64) + np.array([0.9, 0.05, -0.4]) * f)
        return c
    def ctx(rank, count):
        t = pt.PathTracer(shard_rank=rank, shard_count=count); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **camera(0))); t.resize(w, h); t.set_neeat(True); return t
    one = ctx(0, 1); ranks = [ctx(r, world) for r in range(world)]
    owned = [r.shard_info() for r in ranks]
    assert sum(n for n, _ in owned) == w * h
    for f in range(frames):
        for t in [one] + ranks:
            t.set_camera(scenes.bridge_camera(w, h, **camera(f)))
            if view_projection: t.set_view_projection(scenes.view_projection(w, h, **camera(f)))
        one.render(f, 1)
        for r in ranks: r.render(f, 1)
        bufs = []
        for r, (n, _) in zip(ranks, owned):
            b = torch.empty((n, 3), dtype=torch.int32, device="cuda"); r.neeat_pack_feedback(b.data_ptr(), 12 * n); bufs.append(b)
        for i, r in enumerate(ranks):
            for j in range(world):
                if j != i: r.neeat_unpack_feedback(bufs[j].data_ptr(), 12 * owned[j][0], j)
        t1, j1 = one.neeat_tables(); w1, c1 = one.light_feedback(0); p1 = one.lights()["proxyCounters"]
        for i, r in enumerate(ranks):
            tr, jr = r.neeat_tables(); wr, cr = r.light_feedback(0)
            assert jr == j1 and np.array_equal(tr, t1) and np.array_equal(r.lights()["proxyCounters"], p1), "frame %d rank %d: tables / counters" % (f, i)
            assert np.array_equal(wr.view(np.uint32), w1.view(np.uint32)) and np.array_equal(cr, c1), "frame %d rank %d: reservoirs" % (f, i)
    for j in range(1, world):            # the frame gather: ranks 1.. into rank 0
        n, nbytes = owned[j]; b = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        ranks[j].pack_shard(b.data_ptr(), nbytes); ranks[0].unpack_shard(b.data_ptr(), nbytes, j)
    assert np.array_equal(ranks[0].radiance().view(np.uint32), one.radiance().view(np.uint32))
    for t in ranks + [one]: t.close()
