"""Generates tests/golden/neeat_loop_4k_golden.npz: REFERENCE MODE WITH THE REFERENCE'S DEFAULT SAMPLER at full size through the REFERENCE'S text — three accumulated frames of the bench
scene at 3840x2160 with NEE-AT's light baker in the loop (every frame: LightsBaker.hlsl's feedback passes thread by thread, then one sample of PathTracer.hlsli & co.; the host's
world-to-clip matrix set, so the path tracer exports depth, the baker's Reproject tests it and the frustum importance boost weighs the lights): per frame SHA-256 digests of the tile
tables, proxy counters and reservoirs, the tile jitter; the digest of the accumulated frame; the ray counts. tests/test_gpu_full_size.py runs pt_set_neeat + pt_render and compares.
Run in the build container only (about a quarter of an hour of CPU time):   python tests/golden/make_neeat_loop_4k_golden.py"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from rtxpt_amd import scenes
from oracle import ptref
import make_realtime_4k_golden as rt

W, H, FRAMES = rt.W, rt.H, 3
OPTS = dict(global_feedback_weight=0.75, ratio=0.65, ssc_threshold=0.3, prefilter=True)


def workload():
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
    return sc, cam, scenes.default_settings(useFp16Types=1, NEEType=2)


def setup(t, cam):
    M = scenes.view_projection(W, H, **cam)
    t.set_view_projection(M); t.set_light_importance_boost(M); t.set_neeat(True, **OPTS)


if __name__ == "__main__":
    sc, cam, S = workload()
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=True)
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(W, H, **cam)); o.set_settings(S); o.resize(W, H); setup(o, cam)
    out = {}
    for f in range(FRAMES):
        t0 = time.time(); o.render(f, 1)
        tab, jit, cnt = o.neeat_tables(); fw, fc = o.light_feedback(0)
        out["table%d" % f] = rt.digest(tab); out["jitter%d" % f] = np.array(jit, np.uint32); out["counters%d" % f] = rt.digest(cnt); out["fbw%d" % f] = rt.digest(fw); out["fbc%d" % f] = rt.digest(fc)
        print("frame %d: %.0f s, reservoirs filled %d" % (f, time.time() - t0, int((fw > 0).sum())), flush=True)
    c = o.counters()
    out["frame"] = rt.digest(o.radiance()); out["rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "neeat_loop_4k_golden.npz"), **out)
    print("rays", out["rays"].tolist())
