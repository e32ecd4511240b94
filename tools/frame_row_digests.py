"""developer tool (GPU box): per-row SHA-256 of a configuration's frame as the device renders it (tests/golden/make_config_frames_golden.py's definitions) -> gpurun_out/<name>_row_digests.npy,
for hunting a digest mismatch down to its rows.  usage: python tools/frame_row_digests.py C4 [spp-split ...]"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import rtxpt_amd as pt
from rtxpt_amd import scenes
import make_config_frames_golden as cg

name = sys.argv[1]
make, S, w, h, first, n, t = cg.configs()[name]
sc, cam = make()
g = pt.PathTracer(); g.set_scene(sc); g.set_camera(scenes.bridge_camera(w, h, **cam)); g.set_settings(S); g.resize(w, h)
if t is not None: g.animate(scenes.animate_instances(sc, t), scenes.animate_positions(sc, t)); g.reset_accumulation()
splits = [int(x) for x in sys.argv[2:]] or [n]
assert sum(splits) == n
s0 = first
for k in splits: g.render(s0, k); s0 += k
frame = g.radiance()
rows = np.stack([np.frombuffer(hashlib.sha256(np.ascontiguousarray(frame[y]).tobytes()).digest(), np.uint8) for y in range(h)])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "%s_row_digests_%s.npy" % (name, "_".join(str(k) for k in splits))), rows)
print(name, splits, "whole frame", hashlib.sha256(np.ascontiguousarray(frame, np.float32).tobytes()).hexdigest()[:16])
