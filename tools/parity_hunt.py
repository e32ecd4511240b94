"""developer tool: who is right where the 4K C3 frame of the HIP path and the oracle differ?
Renders the frame with both GPU builders (PLOC, Karras: the closest hit is BVH-independent by contract, so the two frames must be identical),
compares a block with the oracle's BVH, and re-renders every differing pixel with the oracle in brute-force mode (every ray tests every triangle:
the definition all three BVHs have to reproduce)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtxpt_amd as pt
from rtxpt_amd import scenes
from oracle import ptref

W, H, SPP = 3840, 2160, 4
rect = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (W // 2 - 480, H // 2 - 270, W // 2 + 480, H // 2 + 270)
sc, cam = scenes.bistro_like(scale=1.0, tex_size=1024)
S = scenes.default_settings()
camd = scenes.bridge_camera(W, H, **cam)
frames = {}
for builder in ("ploc", "karras"):
    os.environ["MI355PT_BVH_BUILDER"] = builder
    g = pt.PathTracer(); g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(W, H)
    st = g.render(0, SPP); frames[builder] = g.radiance()[..., :3].copy()
    print(builder, "rays", st["extendRays"], st["shadowRays"], "ms", st["gpuMilliseconds"]); g.close()
d = (frames["ploc"].view(np.uint32) != frames["karras"].view(np.uint32)).any(-1)
print("GPU ploc vs GPU karras: %d differing pixels of %d" % (int(d.sum()), d.size))
for (y, x) in np.argwhere(d)[:20]:
    print("   pixel", x, y, frames["ploc"][y, x], frames["karras"][y, x])
o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(W, H)
t0 = time.time(); o.render(0, SPP, rect=rect); print("oracle block %s in %.1f s" % (rect, time.time() - t0))
ob = o.radiance()[rect[1]:rect[3], rect[0]:rect[2], :3].copy()
for builder in ("ploc", "karras"):
    gb = frames[builder][rect[1]:rect[3], rect[0]:rect[2]]
    dd = (gb.view(np.uint32) != ob.view(np.uint32)).any(-1)
    print("GPU %s vs oracle BVH: %d differing pixels of %d" % (builder, int(dd.sum()), dd.size))
    bad = np.argwhere(dd)[:24]
    for (yy, xx) in bad:
        x, y = int(rect[0] + xx), int(rect[1] + yy)
        o.set_brute_force(True); o.reset_accumulation(); o.render(0, SPP, rect=(x, y, x + 1, y + 1)); bf = o.radiance()[y, x, :3].copy()
        o.set_brute_force(False)
        print("   pixel (%d,%d): gpu %s oracle-bvh %s brute %s -> gpu==brute %s, oracle-bvh==brute %s" % (x, y, gb[yy, xx], ob[yy, xx], bf,
              np.array_equal(gb[yy, xx].view(np.uint32), bf.view(np.uint32)), np.array_equal(ob[yy, xx].view(np.uint32), bf.view(np.uint32))))
