"""Generates tests/golden/fuzz_sp_hd_golden.npz: 30 seeded random stable-plane frames (tests/fuzz_cases.stable_planes_case: the delta-tree zoo or a small street scene from random
viewpoints; random plane counts, vertex depths, primary-surface replacement, bounce limits, nested qualities, roulette, firefly thresholds, NEE on / off, lp build, previous poses,
1-3 fill sub-samples) at 1280x720 through the REFERENCE'S text of both passes: digests of every plane buffer and of the live plane records after the build pass and after the fill
passes, ray counts. tests/test_gpu_parity_hd.py compares the device with it. Run in the build container only (a few minutes):   python tests/golden/make_fuzz_sp_hd_golden.py"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import ptref
import fuzz_cases as fz
import make_stable_planes_hd_golden as sph


def digests(frame):
    w, h = sph.W, sph.H; sph.W, sph.H = fz.W, fz.H
    try: return sph.digests(frame)
    finally: sph.W, sph.H = w, h


if __name__ == "__main__":
    out = {}
    for seed in fz.SP_SEEDS:
        sc, camd, S, prm, lp16, prev_pose, sample, subs = fz.stable_planes_case(seed)
        t0 = time.time()
        b = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16, mode=1); b.set_scene(sc); b.set_camera(camd); b.set_settings(S); b.resize(fz.W, fz.H)
        if prev_pose is not None: b.set_previous_pose(*prev_pose)
        frame = b.build_stable_planes(sample, prm); out["%d_build_rays" % seed] = np.array([b.counters()["extendRays"]], np.uint64)
        for k, v in digests(frame).items(): out["%d_build_%s" % (seed, k)] = v
        f = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16, mode=2); f.set_scene(sc); f.set_camera(camd); f.set_settings(S); f.resize(fz.W, fz.H)
        for s in range(subs): f.fill_stable_planes(sample + s, prm, frame)
        c = f.counters(); out["%d_fill_rays" % seed] = np.array([c["extendRays"], c["shadowRays"]], np.uint64)
        for k, v in digests(frame).items(): out["%d_fill_%s" % (seed, k)] = v
        hd = frame["header"]
        print("%d: planes %s build rays %d fill rays %s subs %d lp16 %s  %.0f s" % (seed, [int((hd[i] != 0xFFFFFFFF).sum()) for i in range(3)], int(out["%d_build_rays" % seed][0]), out["%d_fill_rays" % seed].tolist(), subs, lp16, time.time() - t0), flush=True)
        b.close(); f.close()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_sp_hd_golden.npz"), **out)
