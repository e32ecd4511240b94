"""Generates tests/golden/bench_frame_golden.npz: THE FRAME bench.py TIMES — its default workload, BASELINE.json's C3 (bistro-like scene at full size: 2.8 M triangles, 64 materials,
32 textures of 1024^2, 7 368 lights, 2048^2 BC6H environment cube; 3840x2160, 4 samples from index 0, 8 bounces, the reference's default lp16 build) — rendered by the REFERENCE'S
integrator text (PathTracer.hlsli & co. compiled from /root/reference by oracle/refpin/hlsl_tu.py --integrator over the oracle's scene services), kept as a SHA-256 of the whole RGBA32F
frame, every 120th row and the ray counts. bench.py compares the device's frame with it (parity.reference_text), tests/test_gpu_full_size.py does the same, tests/test_bench_frame_golden.py
checks the oracle against the kept rows on the CPU. Run in the build container only (the GPU box has no /root/reference; minutes of CPU time):
    python tests/golden/make_bench_frame_golden.py"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes

W, H, SPP, ROW_STEP = 3840, 2160, 4, 120


def bench_workload():
    """bench.py's default scene, camera and settings (keep in step with bench.py main())"""
    sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
    sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
    S = scenes.default_settings(useFp16Types=1)
    return sc, cam, S


if __name__ == "__main__":
    sc, cam, S = bench_workload()
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=True)
    o.set_scene(sc); o.set_camera(scenes.bridge_camera(W, H, **cam)); o.set_settings(S); o.resize(W, H)
    t0 = time.time(); o.render(0, SPP); dt = time.time() - t0
    rad = o.radiance(); c = o.counters()
    out = {"sha256": pin_scenes.frame_digest(rad), "rows": rad[::ROW_STEP].copy(), "row_step": np.array([ROW_STEP], np.uint32),
           "rays": np.array([c["extendRays"], c["shadowRays"]], np.uint64), "size": np.array([W, H, SPP], np.uint32)}
    print("reference text: %dx%d x %d spp, rays %s, %.0f s" % (W, H, SPP, out["rays"].tolist(), dt))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_frame_golden.npz"), **out)
