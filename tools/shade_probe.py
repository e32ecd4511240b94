"""developer probe: per-kernel ms of one C3 frame under settings variants (which part of k_shade costs what). usage: python tools/shade_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtxpt_amd as pt
from rtxpt_amd import scenes

W, H, SPP = 3840, 2160, 4
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
camd = scenes.bridge_camera(W, H, **cam)
g = pt.PathTracer(device=0)
g.set_scene(sc); g.set_camera(camd); g.resize(W, H)
variants = {"default": {}, "nee_off": {"NEEEnabled": 0}, "cand1": {"NEECandidateSamples": 1}, "bounce1": {"bounceCount": 1, "diffuseBounceCount": 1},
            "no_rr": {"enableRussianRoulette": 0}, "uniform_nee": {"NEEType": 0}}
for name, kw in variants.items():
    g.set_settings(scenes.default_settings(**kw))
    for it in range(2):
        g.reset_accumulation(); st = g.render(0, SPP)
    print("%-12s total %.1f ms | ext %.1f shade %.1f shadow %.1f | extend rays %.1fM shadow rays %.1fM hits %.1fM launches %d" % (
        name, st["gpuMilliseconds"], st["extendKernelMs"], st["shadeKernelMs"], st["shadowKernelMs"], st["extendRays"] / 1e6, st["shadowRays"] / 1e6, st["hits"] / 1e6, st["extendLaunches"]))
