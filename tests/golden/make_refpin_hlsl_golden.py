"""Generates tests/golden/refpin_hlsl_golden.npz: outputs of functions of the reference's .hlsli files, compiled verbatim from
/root/reference through oracle/refpin (hlsl_tu.py + hlsl_shim.h -> oracle/_ref/librefpin_hlsl.so), on the seeded inputs of tests/pin_inputs.py.
Run in the build container only (the GPU box has no /root/reference, hence the committed fixture):   python tests/golden/make_refpin_hlsl_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ptref
import pin_inputs

assert ptref.refpin_hlsl() is not None, "librefpin_hlsl.so needs /root/reference"
out = {}
for fn, name in enumerate(ptref.PIN_NAMES):
    a = pin_inputs.rows(name, 256, 0x5EED0100 + fn)
    out["in_" + name] = a
    out["out_" + name] = ptref.pin_call(fn, a, reference=True)
out["bsdf_in"] = pin_inputs.bsdf_cases(3000, 0x5EED0200)
out["bsdf_out"] = ptref.bsdf_probe(out["bsdf_in"], reference=True)
out["stream_in"] = pin_inputs.stream_cases(4000, 0x5EED0300)
out["stream_out"] = ptref.sample_streams(out["stream_in"], reference=True)
lights = pin_inputs.light_inputs(512, 0x5EED0400, lambda kind, w: ptref.light_probe(kind, w, reference=True))
for kind, words in lights.items():
    out["light%d_in" % kind] = words
    out["light%d_out" % kind] = ptref.light_probe(kind, words, reference=True)
tm = pin_inputs.tonemap_cases(0x5EED0500, ptref.TONEMAP_DTYPE)
out["tonemap_out"] = np.stack([ptref.tonemap_linear(rgba, p, reference=True) for p, rgba in tm])
k0, pyr, k1 = pin_inputs.lightbake_inputs(0x5EED0600, lights[2][:, :12])
out["lightbake0_out"] = ptref.lightbake_probe(0, k0, reference=True)
out["lightbake1_out"] = ptref.lightbake_probe(1, k1, pyr, (0.7, 1.3, 0.9), 0.0002, reference=True)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "refpin_hlsl_golden.npz"), **out)
print("wrote %d functions" % len(ptref.PIN_NAMES))
