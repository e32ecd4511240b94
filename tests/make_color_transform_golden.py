"""Generates tests/golden/color_transform_golden.npz: the tone mapper's colour transform (white balance x exposure) from the reference's own
ColorUtils.h / ToneMappingPasses.cpp text (oracle/_ref/librefpin_mat.so, built from /root/reference by oracle/Makefile). Run in the build container."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import ptref


def cases():
    rng = np.random.default_rng(20260924)
    rows = [(0, 6500.0, 0, 0.0, 100.0, 1.0, 1.0), (1, 6500.0, 0, 0.0, 100.0, 1.0, 1.0), (1, 1667.0, 0, 0.0, 100.0, 1.0, 1.0), (1, 25000.0, 0, 0.0, 100.0, 1.0, 1.0),
            (1, 2222.0, 0, 0.0, 100.0, 1.0, 1.0), (1, 4000.0, 0, 0.0, 100.0, 1.0, 1.0), (1, 3999.9998, 1, 1.5, 100.0, 1.0, 1.0)]
    for _ in range(400):
        rows.append((int(rng.integers(0, 2)), float(np.float32(rng.uniform(1667.0, 25000.0))), int(rng.integers(0, 4) == 0), float(np.float32(rng.uniform(-4, 4))),
                     float(np.float32(rng.uniform(25, 3200))), float(np.float32(rng.uniform(0.001, 100.0))), float(np.float32(rng.uniform(0.7, 22.0)))))
    return np.array(rows, np.float64)


if __name__ == "__main__":
    c = cases()
    out = np.stack([ptref.reference_color_transform(int(r[0]), r[1], int(r[2]), r[3], r[4], r[5], r[6]) for r in c])
    # the whole UI block through ToneMappingPass::PreRender + the constant fill (same generator as tests/test_color_transform.py::_random_ui, same seed)
    import rtxpt_amd as pt
    from test_color_transform import _random_ui, _ui_words
    rng = np.random.default_rng(20260925)
    consts = []
    for _ in range(300):
        u = _random_ui(rng); avg = float(np.float32(rng.uniform(0.01, 4))); en = int(rng.integers(0, 2))
        consts.append(ptref.reference_tonemap_constants(_ui_words(u), avg, en))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "color_transform_golden.npz"), cases=c, transform=out,
                        ui_defaults=ptref.reference_tonemap_defaults(), ui_constants=np.stack(consts))
    print(out.shape, out[1].reshape(3, 3))
