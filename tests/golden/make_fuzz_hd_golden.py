"""Generates tests/golden/fuzz_hd_golden.npz: the 40 seeded random scene / camera / settings cases of tests/fuzz_cases.py in both lp builds at 1280x720 through the REFERENCE'S
integrator text (random bounce limits, candidate and full sample counts, roulette, nested-dielectric qualities, firefly thresholds, LOD bias, samplers, BRDF models, depth of field,
animated poses): SHA-256 of each frame and the ray counts. tests/test_gpu_parity_hd.py compares the device (animated cases through pt_animate: refit or rebuild) with it.
Run in the build container only (about a quarter of an hour of CPU time):   python tests/golden/make_fuzz_hd_golden.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ptref
import fuzz_cases as fz


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


if __name__ == "__main__":
    out = {}
    for seed in fz.SEEDS:
        for lp16 in (False, True):
            sc, camd, S, first, count, pose = fz.case(seed, lp16)
            if pose is not None: sc = dict(sc); sc["instances"], sc["positions"] = pose
            t0 = time.time()
            o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(fz.W, fz.H); o.render(first, count)
            c = o.counters(); key = "%d_%s" % (seed, "lp16" if lp16 else "fp32")
            out[key] = digest(o.radiance()); out[key + "_rays"] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
            print("%-10s samples %d rays %s %.0f s" % (key, count, out[key + "_rays"].tolist(), time.time() - t0), flush=True); o.close()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_hd_golden.npz"), **out)
