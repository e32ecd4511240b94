"""developer tool: world-space triangles of a generated scene -> flat float32 file for tools/bvh_lab/bvh_lab.cpp (9 floats per triangle + 1 flag word)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtxpt_amd import scenes

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/bvh_lab_tris.bin"
sc, cam = scenes.bistro_like(scale=scale, tex_size=16)
P, I = sc["positions"], sc["indices"]
tris = []
for inst in sc["instances"]:
    T = inst["transform"].reshape(3, 4)
    m = sc["meshes"][inst["meshIndex"]]
    for gi in range(m["firstGeometry"], m["firstGeometry"] + m["numGeometries"]):
        g = sc["geometries"][gi]
        idx = I[g["indexOffset"]:g["indexOffset"] + g["numIndices"]].astype(np.int64) + int(g["vertexOffset"])
        p = P[idx] @ T[:, :3].T + T[:, 3]
        tris.append(p.reshape(-1, 9).astype(np.float32))
tris = np.concatenate(tris)
print("triangles", tris.shape[0], "camera", cam)
with open(out, "wb") as f:
    f.write(np.array([tris.shape[0]], np.uint32).tobytes())
    f.write(tris.tobytes())
