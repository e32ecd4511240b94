"""Generates tests/golden/stable_planes_golden.npz: what the REFERENCE TEXT of the realtime mode's pre-pass writes (PathTracer.hlsli, PathTracerStablePlanes.hlsli, StablePlanes.hlsli,
the Bridge's motion-vector / guide-buffer exports and postProcessHit of PathTracerSample.hlsl, compiled with PATH_TRACER_MODE_BUILD_STABLE_PLANES by oracle/refpin/hlsl_tu.py --integrator)
for the cases of tests/stable_planes_cases.py: the header (branch ids, first-hit length | dominant plane), the 80-byte records of the planes that exist, stable radiance, depth, motion vectors
and throughput. The CPU tests compare the oracle with it, the GPU tests the device — with no oracle code in the loop.
Run in the build container only (the GPU box has no /root/reference):   python tests/golden/make_stable_planes_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ptref
import stable_planes_cases as spc

out = {}
for name in spc.cases():
    sc, camd, S, prm, lp16 = spc.setup(name)
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16, mode=1)
    o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(spc.W, spc.H)
    r = o.build_stable_planes(spc.SAMPLE, prm)
    for k in spc.KEYS:
        if k != "planes": out[name + "_" + k] = r[k].copy()
    out[name + "_live_planes"] = spc.live_planes(r)
    # the noisy passes over that frame (PATH_TRACER_MODE_FILL_STABLE_PLANES, the other pin library): the planes' noisy radiance | specular average and the specular hit distance
    f = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16, mode=2)
    f.set_scene(sc); f.set_camera(camd); f.set_settings(S); f.resize(spc.W, spc.H)
    for s in range(spc.SUBSAMPLES): f.fill_stable_planes(spc.SAMPLE + s, prm, r)
    lp = spc.live_planes(r)
    assert np.array_equal(np.delete(lp, (16, 17), 1), np.delete(out[name + "_live_planes"], (16, 17), 1)) and np.array_equal(r["header"], out[name + "_header"])      # the pass writes nothing else
    out[name + "_fill_noisy"] = lp[:, 16:18].copy(); out[name + "_fill_spec_hit_t"] = r["spec_hit_t"]
    if name == "zoo_fp32": out[name + "_fill_merge"] = ptref.stable_planes_merge(r, reference=True)      # PostProcess.hlsl NO_DENOISER_FINAL_MERGE (GetAllRadiance of StablePlanes.hlsli)
    if name == "zoo_fp32": out[name + "_fill_spec_hit_t_denoised"] = ptref.denoise_spec_hit_t(r["depth"], r["spec_hit_t"], reference=True)      # DenoisingGuidesBaker.hlsl's DenoiseSpecHitT, ping + pong
    out[name + "_fill_rays"] = np.array([f.counters()["extendRays"], f.counters()["shadowRays"]], np.uint64)
    print("   fill:", out[name + "_fill_rays"].tolist(), "planes with noisy radiance", int((out[name + "_fill_noisy"] != 0).any(-1).sum())); f.close()
    hd = r["header"]
    print(name, "planes", [int((hd[i] != 0xFFFFFFFF).sum()) for i in range(3)], "dominant", np.unique(hd[3] & 3, return_counts=True)[1].tolist(), "rays", o.counters()["extendRays"])
    o.close()
# object motion: the same frames with a previous pose in the scene (InstanceData.prevTransform, GeometryData.prevPositionOffset of the pin's bindings) — the build pass only, it is the one that asks
from rtxpt_amd import scenes
for name, base in spc.motion_cases().items():
    sc, camd, S, prm, lp16 = spc.setup(base)
    o = ptref.Oracle(reference_integrator=True, settings=S, lp16=lp16, mode=1)
    o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(spc.W, spc.H); o.set_previous_pose(*scenes.previous_pose(sc))
    r = o.build_stable_planes(spc.SAMPLE, prm)
    for k in spc.KEYS:
        if k != "planes": out[name + "_" + k] = r[k].copy()
    out[name + "_live_planes"] = spc.live_planes(r)
    moved = (out[name + "_motion_vectors"] != out[base + "_motion_vectors"]).reshape(spc.H, spc.W, -1).any(-1)
    print(name, "pixels whose motion vectors differ from", base, ":", int(moved.sum()), "of", moved.size); assert moved.sum() > moved.size // 2
    o.close()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stable_planes_golden.npz"), **out)
