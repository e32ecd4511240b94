import sys, time, numpy as np
sys.path.insert(0, '.')
import rtxpt_amd as pt
from rtxpt_amd import scenes
from oracle import ptref

def run(variant, W, H, spp):
    sc, cam = scenes.cornell_box(variant)
    S = scenes.config_settings(variant)
    camd = scenes.bridge_camera(W, H, **cam)
    g = pt.PathTracer()
    g.set_scene(sc); g.set_camera(camd); g.set_settings(S); g.resize(W, H)
    st = g.render(0, spp)
    img = g.radiance()
    print(variant, 'gpu stats', st, g.build_stats(), g.scene_info())
    o = ptref.Oracle(); o.set_scene(sc); o.set_camera(camd); o.set_settings(S); o.resize(W, H)
    t = time.time(); o.render(0, spp); print('oracle time', time.time() - t, o.counters())
    ref = o.radiance()
    d = img[..., :3] - ref[..., :3]
    print(variant, 'relL2', np.linalg.norm(d) / np.linalg.norm(ref[..., :3]), 'max abs', np.abs(d).max(), 'nonidentical px', int((np.abs(d).max(-1) > 0).sum()), 'of', W * H,
          'mean gpu', img[..., :3].mean((0, 1)), 'mean ref', ref[..., :3].mean((0, 1)))
    lg, lo = g.lights(), o.lights()
    for k in lg:
        if isinstance(lg[k], np.ndarray): print('  lights', k, lg[k].shape, 'equal' if np.array_equal(lg[k], lo[k]) else 'DIFF %d' % int((lg[k] != lo[k]).sum()))
    return img, ref

run('C1', 256, 256, 1)
run('C2', 320, 180, 2)
