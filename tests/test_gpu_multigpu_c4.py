"""The multi-GPU row on the hardware a one-GPU box offers (run with -m gpu):
  * BASELINE configs[3] — C4: 3840x2160, 16 spp, pixel tiles sharded 8 ways — what ranks 0 and 7 of 8 render is compared with the oracle on complete
    pixel rows of their own tiles, bit for bit; and all 8 shards, rendered one after the other on this GPU and carried through the library's
    pack / unpack entry points, reassemble to exactly the frame a single rank renders;
  * the frame gather's RCCL path — pt_comm_unique_id -> pt_comm_init(world = 1) -> pt_render -> pt_gather — runs on the device: with a communicator a world
    of one performs the whole protocol as a loop-back (pack, ncclSend to self + ncclRecv from self in one group, unpack; the frame is poisoned in between),
    so the radiance read back afterwards has been through the dlopen'd RCCL.
The real multi-rank run is the driver's (bench.py --gpus 8); the protocol itself is also driven by two and three gloo processes in test_multigpu_cpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 3840, 2160


def _imports():
    import rtxpt_amd as pt
    from rtxpt_amd import scenes, parallel
    from oracle import ptref
    return pt, scenes, parallel, ptref


@pytest.mark.tail_once("tail_default")      # (half a minute of 16-spp frames: in the product's configuration; the tail-off kernels render the same frames in tests/test_gpu_full_size.py)
def test_c4_ranks_0_and_7_of_8_match_the_oracle_at_16_spp():
    pt, scenes, parallel, ptref = _imports()
    SPP, WORLD = 16, 8
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
    S = scenes.default_settings(useFp16Types=1)
    camd = scenes.bridge_camera(W, H, **cam)
    o = ptref.Oracle(lp16=True); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(W, H)
    for rank in (0, 7):
        g = pt.PathTracer(shard_rank=rank, shard_count=WORLD); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(W, H)
        st = g.render(0, SPP); a = g.radiance()
        px = parallel.shard_pixels(W, H, rank, WORLD)
        assert st["pathsTraced"] == px.size * SPP
        own = np.zeros((H, W), bool); own[(px & 0xFFFF).astype(np.int64), (px >> 16).astype(np.int64)] = True
        assert np.all(a[~own] == 0.0), "a rank writes nothing outside its tiles"
        assert np.isfinite(a).all() and (a >= 0).all() and np.all(a[own][:, 3] == 1.0)
        # accumulation over calls is associative at this size too: 16 = 4 + 12
        g.reset_accumulation(); g.render(0, 4); g.render(4, 12)
        assert np.array_equal(a, g.radiance())
        ys = (px & 0xFFFF).astype(np.int64)
        for y in (int(ys[7]), int(ys[ys.size // 2]) + 3, int(ys[-1]) - 5):       # three complete rows through the rank's first, middle and last tile: the oracle renders the row, the comparison takes the rank's pixels of it
            o.reset_accumulation(); o.render(0, SPP, rect=(0, y, W, y + 1))
            want = o.radiance()[y, :, :3]; got = a[y, :, :3]; m = own[y]
            assert m.any()
            bad = int((got[m].view(np.uint32) != want[m].view(np.uint32)).any(-1).sum())
            assert bad == 0, "rank %d row %d: %d of %d owned pixels differ" % (rank, y, bad, int(m.sum()))
        g.close()
    o.close()


@pytest.mark.tail_once("tail_default")      # (half a minute of 16-spp frames: in the product's configuration; the tail-off kernels render the same frames in tests/test_gpu_full_size.py)
def test_c4_eight_shards_reassemble_to_the_single_rank_frame():
    """All 8 ranks' shards of the 4K frame (4 spp here: every rank is rendered in turn on the one GPU), packed by pt_pack_shard and put back by pt_unpack_shard on a
    rank-0 context, equal the frame one rank renders alone."""
    import torch
    pt, scenes, parallel, ptref = _imports()
    SPP, WORLD = 4, 8
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
    S = scenes.default_settings(useFp16Types=1)
    camd = scenes.bridge_camera(W, H, **cam)
    full = pt.PathTracer(); full.set_scene(sc); full.set_camera(camd); full.set_settings(S); full.resize(W, H); full.render(0, SPP)
    want = full.radiance(); full.close()
    packed = []
    root = None
    for rank in range(WORLD):
        g = pt.PathTracer(shard_rank=rank, shard_count=WORLD); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(W, H)
        g.render(0, SPP)
        n, nbytes = g.shard_info()
        assert n == parallel.shard_pixels(W, H, rank, WORLD).size and nbytes == 16 * n
        buf = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        g.pack_shard(buf.data_ptr(), nbytes); packed.append(buf)
        if rank == 0: root = g
        else: g.close()
    for rank in range(1, WORLD):
        root.unpack_shard(packed[rank].data_ptr(), packed[rank].numel() * 4, rank)
    got = root.radiance(); root.close()
    assert np.array_equal(got, want)


def test_gather_runs_through_rccl_on_one_device():
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C2")
    S = scenes.config_settings("C2")
    w, h = 200, 136
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(w, h, **cam)); g.set_settings(S); g.resize(w, h)
    g.render(0, 2); before = g.radiance()
    g.gather()                                                    # no communicator, world 1: nothing to do, the frame is untouched
    assert np.array_equal(before, g.radiance())
    uid = pt.comm_unique_id()
    assert len(bytes(uid)) == pt.COMM_ID_BYTES
    g.comm_init(uid, 0, 1)
    g.gather()                                                    # pack -> ncclSend(self) + ncclRecv(self) -> unpack, the frame poisoned in between
    after = g.radiance()
    assert np.array_equal(before, after)
    g.render(2, 2); g.gather(); g.gather()                        # a second frame on the same communicator; gathering twice changes nothing
    g2 = pt.PathTracer(); g2.set_scene(sc); g2.set_camera(scenes.bridge_camera(w, h, **cam)); g2.set_settings(S); g2.resize(w, h); g2.render(0, 4)
    assert np.array_equal(g.radiance(), g2.radiance())
    g.resize(64, 48); g.render(0, 1); b2 = g.radiance(); g.gather()      # the per-size gather state follows a resize
    assert np.array_equal(b2, g.radiance())
    g.comm_destroy(); g.close(); g2.close()


def test_environment_without_params_renders_the_supplied_radiance():
    """pt_set_environment(ctx, rgb, w, h, NULL): identity orientation and ColorMultiplier = 1 / c_envMapRadianceScale — the image's radiance as supplied
    (the baked cube holds radiance x 1/4, Sample.cpp:88, 1939-1940). A uniform environment seen directly must come back as its own value."""
    import ctypes
    pt, scenes, parallel, ptref = _imports()
    sc, cam = scenes.cornell_box("C2")
    sc = dict(sc); sc["env"] = None
    w, h = 48, 32
    g = pt.PathTracer(); g.set_scene(sc)
    rgb = np.full((16, 32, 3), 0.75, np.float32)
    g._chk(g.L.pt_set_instances(g.h, None, 0), "pt_set_instances")      # an empty scene: every camera ray sees the environment
    g._chk(g.L.pt_set_environment(g.h, rgb.ctypes.data_as(ctypes.c_void_p), 32, 16, None), "pt_set_environment")
    g._chk(g.L.pt_set_environment_bake(g.h, 64, None, 0), "pt_set_environment_bake")
    S = scenes.config_settings("C2"); g.set_camera(scenes.bridge_camera(w, h, **cam)); g.set_settings(S); g.resize(w, h)
    g.render(0, 1); a = g.radiance()
    assert np.allclose(a[..., :3], 0.75, rtol=2e-3), (a[..., :3].min(), a[..., :3].max())      # (binary16 storage of the cube and of the path radiance)
    # the explicit parameters a reference host would pass for tint 1, intensity 1 give the same frame
    I = (ctypes.c_float * 12)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)
    p = pt.PtEnvMapSceneParams(I, (ctypes.c_float * 3)(4.0, 4.0, 4.0), 1.0)
    g._chk(g.L.pt_set_environment(g.h, rgb.ctypes.data_as(ctypes.c_void_p), 32, 16, ctypes.byref(p)), "pt_set_environment")
    g.reset_accumulation(); g.render(0, 1)
    assert np.array_equal(a, g.radiance())
    g.close()
