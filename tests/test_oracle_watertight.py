"""The oracle's hit definition is watertight (CPU; the device against the oracle and at 10^8 rays: tests/test_gpu_watertight.py). oracle/ptref/scene.h intersect_tri_wt — the
Woop-Benthin-Wald test over stored vertices, DXR's promise behind Bridge::traceScatterRay (Rtxpt/Shaders/PathTracerBridgeDonut.hlsli:993-1055) — on a closed, shared-vertex
icosphere under a rotated, non-uniformly scaled transform: rays from inside aimed exactly at shared edges, at vertices, a few fp32 spacings beside an edge and at random surface
points must all hit; and the BVH answer must equal the exhaustive loop over all triangles (the hit definition does not depend on the tree)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import ptref
from rtxpt_amd import scenes
from test_gpu_watertight import _rays


def test_no_ray_escapes_a_closed_mesh_and_the_tree_does_not_matter():
    sc, W, I = scenes.closed_icosphere(4)
    o = ptref.Oracle(); o.set_scene(sc); o.set_settings(scenes.default_settings())
    rng = np.random.default_rng(0x5EED0511)
    for kind, n, off in (("edge", 300_000, 0.0), ("vertex", 150_000, 0.0), ("edge", 100_000, 0.5), ("edge", 100_000, 1.0), ("interior", 300_000, 0.0)):
        rays = _rays(W, I, n, kind, rng, offset_ulps=off)
        hits = o.trace_closest(rays)
        miss = int((hits.view(np.uint32)[:, 1] == 0xFFFFFFFF).sum())
        assert miss == 0, "%d of %d rays aimed at a shared %s (offset %.1f) escape" % (miss, n, kind, off)
        if kind != "interior":      # the exhaustive loop finds the same hit records, bit for bit (4 000 rays: 5 120 triangles each)
            brute = o.trace_closest(rays[:4000], brute=True)
            assert np.array_equal(brute.view(np.uint32), hits[:4000].view(np.uint32)), kind
    o.close()
