"""The realtime mode's coupled frame on the device (run with -m gpu): pt_realtime_frame — LightsBaker::UpdateBegin, the stable-plane build pass, UpdateEnd on that frame's depth and motion
vectors, the fill passes feeding the feedback reservoirs — against the runs the REFERENCE'S TEXT produced (tests/golden/realtime_golden.npz; no oracle code in the loop): every frame's tile
tables, jitter, global proxy counters, reservoirs, noisy radiance, specular hit distances, depth, motion vectors and header, bit for bit, and the ray counts of the whole run."""
import os, sys
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import realtime_cases as rc
from rtxpt_amd import scenes

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "realtime_golden.npz")


def _same(a, b): return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


@pytest.mark.parametrize("name", list(rc.cases()))
def test_device_matches_the_reference_text_run(name):
    import rtxpt_amd as pt
    g = np.load(GOLDEN)
    make, _, w, h, frames, subs, step, kw = rc.cases()[name]; S = rc.settings_for(name)
    sc, cam = make()
    t = pt.PathTracer(); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h); t.set_neeat(True)
    P = rc.poses(sc, kw, frames); kw = {k: v for k, v in kw.items() if k != "anim_dt"}
    if P is not None: t.set_motion_history(True)      # an animated case: every pt_animate is a scene refresh (refit, light re-bake, the pose it finds becomes the previous one)
    rays = [0, 0]
    for f in range(frames):
        if P is not None: t.animate(P[f][0], P[f][1], vertex_ranges=scenes.animated_vertex_ranges(sc) if f % 2 else None)
        cur, prev = rc.camera(cam, step, f), rc.camera(cam, step, max(f - 1, 0))
        prm = scenes.stable_planes_params(w, h, scenes.view_projection(w, h, **cur), prev_world_to_clip=scenes.view_projection(w, h, **prev), sub_samples=subs, **kw)
        t.set_camera(scenes.bridge_camera(w, h, **cur))
        frame, bst, fst = t.realtime_frame(f * subs, prm)
        rays[0] += int(bst["extendRays"]) + int(fst["extendRays"]); rays[1] += int(fst["shadowRays"])
        tab, jit = t.neeat_tables(); fw, fc = t.light_feedback(0)
        got = dict(table=tab, jitter=np.array(jit, np.uint32), counters=t.lights()["proxyCounters"], fbw=fw, fbc=fc, noisy=rc.live_noisy(frame, w, h), spec_hit_t=frame["spec_hit_t"],
                   depth=frame["depth"], motion_vectors=frame["motion_vectors"], header=frame["header"])
        for k, v in got.items():
            want = g["%s_%s%d" % (name, k, f)]
            assert _same(v, want), "%s frame %d: %s differs (%d of %d words)" % (name, f, k, int((np.asarray(v).view(np.uint8) != np.asarray(want).view(np.uint8)).sum()), np.asarray(want).view(np.uint8).size)
    assert rays == [int(x) for x in g[name + "_rays"]]
    t.close()


def test_realtime_frame_without_the_baker_and_refusals():
    """pt_set_neeat off: build + fill with the global sampler == the separate calls, on tile shards too; a shard WITH the baker is refused without a communicator."""
    import rtxpt_amd as pt
    sc, cam = scenes.stable_planes_zoo(); S = scenes.config_settings("C2"); w, h = 64, 48
    prm = scenes.stable_planes_params(w, h, scenes.view_projection(w, h, **cam), sub_samples=2)
    a = pt.PathTracer(); a.set_scene(sc); a.set_settings(S); a.set_camera(scenes.bridge_camera(w, h, **cam)); a.resize(w, h)
    fa, _, _ = a.realtime_frame(3, prm)
    b = pt.PathTracer(); b.set_scene(sc); b.set_settings(S); b.set_camera(scenes.bridge_camera(w, h, **cam)); b.resize(w, h)
    b.build_stable_planes(3, prm); fb = b.fill_stable_planes(3, prm, sub_samples=2)
    for k in ("header", "planes", "spec_hit_t", "depth", "motion_vectors", "stable_radiance", "throughput"): assert _same(fa[k], fb[k]), k
    a.close(); b.close()
    # a tile shard without the baker is just its two passes; with the baker it needs the other ranks' reservoirs, depth and motion vectors: refused without a communicator
    # (test_tile_sharded_realtime_run_with_the_baker_equals_the_unsharded_run drives the parts)
    c = pt.PathTracer(shard_rank=0, shard_count=2); c.set_scene(sc); c.set_settings(S); c.set_camera(scenes.bridge_camera(w, h, **cam)); c.resize(w, h)
    fc, _, _ = c.realtime_frame(3, prm)
    px = pt.shard_layout(w, h, 0, 2); ys, xs = (px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)
    for k in ("depth", "spec_hit_t", "motion_vectors", "stable_radiance", "throughput"): assert _same(fc[k][ys, xs], fa[k][ys, xs]), k
    S2 = S.copy(); S2["NEEType"] = 2; c.set_settings(S2); c.set_neeat(True)
    with pytest.raises(Exception): c.realtime_frame(0, prm)
    c.close()


def test_plane_buffers_of_tile_shards_reassemble():
    """Three ranks of a tile-sharded realtime frame on one device: each builds and fills the planes of its own tiles; packed, carried to rank 0 and unpacked they are the unsharded frame's
    buffers — header, every plane record, stable radiance, depth, hit distances, motion vectors, throughput — and the 5 x 5 fill-in of the hit distances, refused on a lone shard, runs on the
    gathered planes and equals the unsharded result. The RCCL form (pt_gather_stable_planes) runs as a world-of-one loop-back with the buffers poisoned in between."""
    import torch
    import rtxpt_amd as pt
    sc, cam = scenes.stable_planes_zoo(); S = scenes.config_settings("C2"); w, h, world = 96, 72, 3
    prm = scenes.stable_planes_params(w, h, scenes.view_projection(w, h, **cam), sub_samples=1)
    def ctx(rank, count):
        t = pt.PathTracer(shard_rank=rank, shard_count=count); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h); return t
    one = ctx(0, 1); one.build_stable_planes(7, prm); want = one.fill_stable_planes(7, prm)
    ranks = [ctx(r, world) for r in range(world)]
    bufs = []
    for r, t in enumerate(ranks):
        t.build_stable_planes(7, prm); t.fill_stable_planes(7, prm)
        n = t.stable_planes_shard_bytes(r); assert n == t.shard_info()[0] * 284
        b = torch.empty(n // 4, dtype=torch.int32, device="cuda"); t.pack_stable_planes(b.data_ptr(), n); bufs.append(b)
    with pytest.raises(Exception): ranks[0].denoise_spec_hit_t()
    for r in range(1, world): ranks[0].unpack_stable_planes(bufs[r].data_ptr(), bufs[r].numel() * 4, r)
    got = ranks[0].get_stable_planes()
    for k in ("header", "stable_radiance", "depth", "spec_hit_t", "motion_vectors", "throughput"): assert _same(got[k], want[k]), k
    assert _same(rc.live_noisy(got, w, h), rc.live_noisy(want, w, h))
    hd = want["header"]
    for pl in range(3):
        ys, xs = np.nonzero(hd[pl] != 0xFFFFFFFF)
        idx = [scenes.stable_planes_address(x, y, pl, w, h) for x, y in zip(xs.tolist(), ys.tolist())]
        assert _same(got["planes"][idx], want["planes"][idx]), pl
    assert _same(ranks[0].denoise_spec_hit_t(), one.denoise_spec_hit_t())
    # RCCL loop-back on the unsharded context
    uid = pt.comm_unique_id(); one.comm_init(uid, 0, 1)
    before = one.get_stable_planes(); one.gather_stable_planes(); after = one.get_stable_planes()
    for k in ("header", "planes", "stable_radiance", "depth", "spec_hit_t", "motion_vectors", "throughput"): assert _same(before[k], after[k]), k
    one.comm_destroy()
    for t in ranks + [one]: t.close()


def test_tile_sharded_realtime_run_with_the_baker_equals_the_unsharded_run():
    """Three ranks of a tile-sharded realtime run with NEE-AT on one device, the host moving the packed buffers (the RCCL form needs three GPUs): before UpdateBegin every rank
    receives the others' reservoirs, after the build pass their depth and motion vectors, and then runs the baker's passes on the same planes as everybody else. Every frame: the
    tile tables, jitter and proxy counters of every rank equal the unsharded run's; the plane buffers, gathered on rank 0, and the reservoirs after the fill pass equal it too.
    The unsharded run goes through pt_realtime_frame, the ranks through the parts — so the parts == the one call as well."""
    import torch
    import rtxpt_amd as pt
    name = "bistro_like_realtime"
    make, _, w, h, frames, subs, step, kw = rc.cases()[name]; S = rc.settings_for(name); world = 3
    sc, cam = make()
    def ctx(rank, count):
        t = pt.PathTracer(shard_rank=rank, shard_count=count); t.set_scene(sc); t.set_settings(S); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.resize(w, h); t.set_neeat(True); return t
    one = ctx(0, 1); ranks = [ctx(r, world) for r in range(world)]
    def exchange(pack, unpack, bytes_per_pixel):
        bufs = []
        for r, t in enumerate(ranks):
            n = t.shard_info()[0] * bytes_per_pixel; b = torch.empty(n // 4, dtype=torch.int32, device="cuda"); pack(t, b.data_ptr(), n); bufs.append(b)
        for r, t in enumerate(ranks):
            for q in range(world):
                if q != r: unpack(t, bufs[q].data_ptr(), bufs[q].numel() * 4, q)
    for f in range(frames):
        cur, prev = rc.camera(cam, step, f), rc.camera(cam, step, max(f - 1, 0))
        prm = scenes.stable_planes_params(w, h, scenes.view_projection(w, h, **cur), prev_world_to_clip=scenes.view_projection(w, h, **prev), sub_samples=subs, **kw)
        camd = scenes.bridge_camera(w, h, **cur)
        one.set_camera(camd); want, _, _ = one.realtime_frame(f * subs, prm)
        wtab, wjit = one.neeat_tables(); wfw, wfc = one.light_feedback(0); wcnt = one.lights()["proxyCounters"]
        for t in ranks: t.set_camera(camd)
        if f: exchange(lambda t, p, n: t.neeat_pack_feedback(p, n), lambda t, p, n, q: t.neeat_unpack_feedback(p, n, q), 12)
        for t in ranks: t.neeat_update_begin(); t.build_stable_planes(f * subs, prm)
        exchange(lambda t, p, n: t.pack_stable_plane_guides(p, n), lambda t, p, n, q: t.unpack_stable_plane_guides(p, n, q), 16)
        for t in ranks: t.neeat_update_end()
        for r, t in enumerate(ranks):
            tab, jit = t.neeat_tables()
            assert _same(tab, wtab) and tuple(jit) == tuple(wjit) and _same(t.lights()["proxyCounters"], wcnt), "frame %d rank %d: the baker's outputs differ from the unsharded run" % (f, r)
        for t in ranks: t.fill_stable_planes(f * subs, prm, sub_samples=subs)
        # the frame: every rank's plane records to rank 0
        bufs = []
        for r, t in enumerate(ranks):
            n = t.stable_planes_shard_bytes(r); b = torch.empty(n // 4, dtype=torch.int32, device="cuda"); t.pack_stable_planes(b.data_ptr(), n); bufs.append(b)
        for r in range(1, world): ranks[0].unpack_stable_planes(bufs[r].data_ptr(), bufs[r].numel() * 4, r)
        got = ranks[0].get_stable_planes()
        for k in ("header", "stable_radiance", "depth", "spec_hit_t", "motion_vectors", "throughput"): assert _same(got[k], want[k]), "frame %d: %s" % (f, k)
        assert _same(rc.live_noisy(got, w, h), rc.live_noisy(want, w, h)), "frame %d: noisy radiance" % f
        # the reservoirs: every rank holds its own pixels' (the others' arrive with the next frame's exchange)
        fw = np.zeros((h, w), np.float32); fc = np.zeros((h, w), np.uint32)
        for r, t in enumerate(ranks):
            a, b = t.light_feedback(0); px = pt.shard_layout(w, h, r, world); ys, xs = (px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)
            fw[ys, xs] = a[ys, xs]; fc[ys, xs] = b[ys, xs]
        assert _same(fw, wfw) and _same(fc, wfc), "frame %d: reservoirs" % f
    # and what pt_realtime_frame refuses on a lone shard without a communicator
    with pytest.raises(Exception): ranks[1].realtime_frame(99, prm)
    for t in ranks + [one]: t.close()
