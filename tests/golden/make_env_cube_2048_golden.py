"""Generates tests/golden/env_cube_2048_golden.npz: the environment cube at EnvMapBaker's resolution for an image source (2048^2 x 6 + mips: 33.5 M texels) baked by the REFERENCE'S
text (EnvMapBaker.hlsl BaseLayerCS / MIPReduceCS, the directional lights' discs, the BC6U encoder + BC6H decode of its D3D12 build) for four image set-ups — the bench scene's sky as the
bench bakes it ("Fast" compression), a noisy HDR source without compression (every bilinear tap matters), sun discs with "Fast" and with "Quality" compression — and the procedural sky at its 1024^2 for five presets — kept as SHA-256 of
the whole cube. tests/test_gpu_parity_hd.py compares the device's cubes with it. Run in the build container only (a few minutes):  python tests/golden/make_env_cube_2048_golden.py"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rtxpt_amd import scenes
from oracle import ptref
import pin_scenes


def digest(a): return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def cases():
    small = lambda: scenes.bistro_like(scale=0.02, tex_size=128)
    def bench_sky():
        sc, cam = small(); sc = dict(sc); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1; return sc
    def noisy():
        sc, cam = small(); sc = dict(sc); rgb, tw, cm = sc["env"]
        rng = np.random.default_rng(11); src = (rng.random((256, 512, 3), np.float32) ** 4 * 40.0).astype(np.float32)
        sc["env"] = (src, tw, cm); sc["env_cube_dim"] = 2048; sc["env_compression"] = 0; return sc
    def discs(comp): return lambda: pin_scenes.with_sun_discs(small, cube_dim=2048, compression=comp)()[0]
    out = {"bench_sky_fast": bench_sky, "noisy_uncompressed": noisy, "sun_discs_fast": discs(1), "sun_discs_quality": discs(2)}
    # the procedural sky (SampleProceduralSky.hlsli / precomputed_sky.hlsli: precomputed atmosphere, sun disc, ray-marched clouds, the half-resolution cloud pre-pass) at the
    # reference's resolution for a sky, 1024^2 (EnvMapBaker.cpp:374-375): five presets / times of day, two of them over an image source, one compressed
    import test_procedural_sky as tps
    for key, (preset, tm, image, kw) in {"sky_midday": ("==PROCEDURAL_SKY_MIDDAY==", 0.0, False, {}), "sky_evening": ("==PROCEDURAL_SKY_EVENING==", 7.0, False, {}),
                                         "sky_dawn_over_image_fast": ("==PROCEDURAL_SKY_DAWN==", 0.0, True, {"env_compression": 1}), "sky_time_of_day": ("==PROCEDURAL_SKY==", 70000.0, False, {}),
                                         "sky_morning_over_image": ("==PROCEDURAL_SKY_MORNING==", 3.0, True, {})}.items():
        out[key] = (lambda preset=preset, tm=tm, image=image, kw=kw: tps._scene(preset, tm, 1024, image, **kw)[0])
    return out


if __name__ == "__main__":
    out = {}; only = sys.argv[1:]
    for name, make in cases().items():
        if only and not any(o in name for o in only): continue
        sc = make()
        o = ptref.Oracle(reference_integrator=True, settings=scenes.default_settings()); o.set_scene(sc)
        t0 = time.time(); cube, dim, lv = o.env_cube(reference=True)
        out[name] = digest(cube); out[name + "_dim"] = np.array([dim, lv, cube.shape[0]], np.uint32)
        print("%-22s dim %d levels %d texels %d  %.0f s" % (name, dim, lv, cube.shape[0], time.time() - t0), flush=True); o.close()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "env_cube_2048_golden.npz")
    if only and os.path.exists(path): old = dict(np.load(path)); old.update(out); out = old
    np.savez_compressed(path, **out)
