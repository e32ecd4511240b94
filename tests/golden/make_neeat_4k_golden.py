"""Generates tests/golden/neeat_4k_golden.npz: NEE-AT's path-tracer side (NEEType 2, the reference's default sampler: tile-local candidates, two-sampler MIS, temporal feedback
reservoirs) at FULL size through the REFERENCE'S integrator text — the bench scene, 3840x2160, 2 samples, the reference's default lp16 build, synthetic tile tables
(scenes.synthetic_local_light_tables: what the baker would hand over; the baker's own text runs thread by thread and is pinned at small sizes, tests/test_neeat_baker.py): SHA-256 of
the frame and of every sample's reservoir planes, ray counts, the number of baked lights the tables were drawn for. tests/test_gpu_full_size.py compares the device with it.
Run in the build container only (a few minutes of CPU time):   python tests/golden/make_neeat_4k_golden.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref

W, H, FIRST, N = 3840, 2160, 2, 2
OPTS = dict(seed=5, jitter=(3, 5), ratio=0.65, ssc_threshold=0.3, feedback=True)


def workload():
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
    return sc, cam, scenes.default_settings(useFp16Types=1, NEEType=2)


def table(num_lights): return scenes.synthetic_local_light_tables(num_lights, W, H, seed=OPTS["seed"], jitter=OPTS["jitter"])


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


if __name__ == "__main__":
    sc, cam, S = workload()
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=True)
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(W, H, **cam)); o.set_settings(S); o.resize(W, H); o.L.ptref_prepare(o.h)
    n_lights = len(o.lights()["lights"])
    o.set_local_light_sampling(table(n_lights), jitter=OPTS["jitter"], ratio=OPTS["ratio"], ssc_threshold=OPTS["ssc_threshold"], feedback=OPTS["feedback"])
    t0 = time.time(); o.render(FIRST, N); c = o.counters()
    out = {"frame": digest(o.radiance()), "rays": np.array([c["extendRays"], c["shadowRays"]], np.uint64), "lights": np.array([n_lights], np.uint32)}
    for s in range(N):
        wgt, cand = o.light_feedback(s); out["fbw%d" % s] = digest(wgt); out["fbc%d" % s] = digest(cand); print("sample", s, "reservoirs filled", int((wgt > 0).sum()))
    print("rays %s, %.0f s" % (out["rays"].tolist(), time.time() - t0))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "neeat_4k_golden.npz"), **out)
