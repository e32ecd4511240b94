"""developer A/B: for every variant library under gpurun_ab/ (one process each), the bench workload's serial-kernel times per step and the pipelined frame time.
usage: python tools/ab_kernels.py [lib ...]      (no arguments: gpurun_ab/lib_*.so)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import rtxpt_amd as pt
from rtxpt_amd import scenes
W, H, SPP = 3840, 2160, 4
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024); sc["env_cube_dim"] = 2048; sc["env_compression"] = 1
g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(W, H, **cam)); g.set_settings(scenes.default_settings(useFp16Types=1)); g.resize(W, H)
g.reset_accumulation(); g.render(0, SPP)
ms = []
for _ in range(3):
    g.reset_accumulation(); st = g.render(0, SPP); ms.append(st["gpuMilliseconds"])
rays = st["extendRays"] + st["shadowRays"]
g.set_serial_kernels(True); g.reset_accumulation(); g.render(0, SPP); g.reset_accumulation(); os.environ["MI355PT_PASS_LOG"] = "1"; s = g.render(0, SPP); del os.environ["MI355PT_PASS_LOG"]
chk = float(np.float64(g.radiance()[..., :3]).sum())
bi = g.bvh_info()
print("frame %%.2f ms %%.1f Mrays/s | serial: frame %%.2f extend %%.2f shade %%.2f shadow %%.2f | checksum %%.6f | %%s on %%s, build %%.1f ms (host %%.1f), %%d wide nodes" %% (min(ms), rays / min(ms) / 1e3, s["gpuMilliseconds"], s["extendKernelMs"], s["shadeKernelMs"], s["shadowKernelMs"], chk, bi["builderName"], bi["builtOn"], bi["buildMs"], bi["hostMs"], bi["numWideNodes"]))
''' % ROOT
libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "gpurun_ab", "lib_*.so")))
for lib in libs:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, MI355PT_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    out = [l for l in r.stdout.splitlines() if l.startswith("frame")]
    print("%-28s %s%s" % (os.path.basename(lib), (os.environ.get("AB_LABEL", "") + " ") if os.environ.get("AB_LABEL") else "", out[-1] if out else ("FAILED: " + r.stderr[-300:])), flush=True)
    if os.environ.get("AB_PASSES"):
        for l in [l for l in r.stderr.splitlines() if l.startswith("[pass log]   ")][:int(os.environ["AB_PASSES"])]: print("      " + l[13:], flush=True)
