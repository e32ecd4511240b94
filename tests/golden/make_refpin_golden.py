"""Generates tests/golden/refpin_golden.json from the REFERENCE'S OWN code (oracle/_ref/librefpin.so, compiled from
/root/reference by oracle/Makefile). Run in the build container only: /root/reference does not exist on the GPU box,
which is why the outputs are committed as fixtures.   python tests/golden/make_refpin_golden.py
"""
import ctypes, json, math, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import ptref

R = ptref.refpin()
assert R is not None, "librefpin.so needs /root/reference"
rng = np.random.default_rng(0x5EED0001)
xs = [0, 1, 2, 3, 0xFFFFFFFF, 0x80000000, 0x035F9F29, 0x9e3779b9] + [int(v) for v in rng.integers(0, 2**32, 248, dtype=np.uint64)]
out = {"source": "compiled from /root/reference/Rtxpt/Shaders/PathTracer/Utils/NoiseAndSequences.hlsli (Hash32, Hash32Combine, Hash32ToFloat, SobolC), "
                 "PathTracerShared.h (BridgeCamera, layouts), Utils.hlsli (EvalMIS), PolymorphicLight.h, MaterialPT.h, SubInstanceData.h"}
out["hash32"] = [[x, R.refpin_hash32(x)] for x in xs]
out["hash32_combine"] = [[xs[i], xs[(i * 7 + 3) % len(xs)], R.refpin_hash32_combine(xs[i], xs[(i * 7 + 3) % len(xs)])] for i in range(len(xs))]
R.refpin_hash32_to_float.argtypes = [ctypes.c_uint32]
out["hash32_to_float"] = [[x, float(np.float32(R.refpin_hash32_to_float(x))).hex()] for x in xs]
out["sobol"] = [[x, d, R.refpin_sobol(x, d)] for x in xs[:64] + list(range(64)) for d in range(5)]
cams = []
for k in range(8):
    w, h = int(rng.integers(16, 4096)), int(rng.integers(16, 2160))
    pos = rng.uniform(-10, 10, 3).astype(np.float32); d = rng.normal(size=3).astype(np.float32); up = np.array([0, 1, 0], np.float32)
    fov = float(np.float32(rng.uniform(0.3, 1.8))); near, far = 0.05, 1000.0; focal = float(np.float32(rng.uniform(0.5, 20))); ap = float(np.float32(rng.uniform(0, 0.1)))
    jit = rng.uniform(-0.5, 0.5, 2).astype(np.float32)
    buf = (ctypes.c_uint8 * 112)()
    f3 = lambda v: (ctypes.c_float * 3)(*[float(x) for x in v])
    R.refpin_bridge_camera(ctypes.c_uint32(w), ctypes.c_uint32(h), ctypes.c_float(np.float32(w) / np.float32(h)), f3(pos), f3(d), f3(up), ctypes.c_float(fov), ctypes.c_float(near), ctypes.c_float(far),
                           ctypes.c_float(focal), ctypes.c_float(ap), (ctypes.c_float * 2)(float(jit[0]), float(jit[1])), buf)
    cams.append(dict(w=w, h=h, pos=[float(x) for x in pos], dir=[float(x) for x in d], up=[0.0, 1.0, 0.0], fov=fov, near=near, far=far, focal=focal, aperture=ap,
                     jitter=[float(x) for x in jit], bytes=bytes(buf).hex()))
out["bridge_camera"] = cams
lay = (ctypes.c_uint32 * 32)(); R.refpin_layout(lay)
names = ["sizeof_PathTracerCameraData", "sizeof_PathTracerConstants", "sizeof_PolymorphicLightInfo", "sizeof_PolymorphicLightInfoEx", "sizeof_PTMaterialData", "sizeof_SubInstanceData",
         "offsetof_PTMaterialData_IoR", "offsetof_PTMaterialData_Volume", "offsetof_PTMaterialData_BaseOrDiffuseTextureIndex", "offsetof_Camera_ViewportSize", "offsetof_Camera_Jitter",
         "PTMaterialFlags_ThinSurface", "PTMaterialFlags_UseBaseOrDiffuseTexture", "PTMaterialFlags_UseEmissiveTexture", "PTMaterialFlags_UseNormalTexture",
         "PTMaterialFlags_UseMetalRoughOrSpecularTexture", "PTMaterialFlags_UseTransmissionTexture", "PTMaterialFlags_NestedPriorityShift", "kTriangle", "kEnvironmentQuad",
         "kPolymorphicLightTypeShift", "Flags_AlphaTested", "Flags_ExcludeFromNEE", "PATH_TRACER_MAX_PAYLOAD_SIZE"]
out["layout"] = {n: int(lay[i]) for i, n in enumerate(names)}
mis = []
for k in range(32):
    n0, p0, n1, p1 = [float(np.float32(v)) for v in (rng.integers(1, 6), rng.uniform(0, 50), rng.integers(1, 6), rng.uniform(0, 50))]
    mis.append([n0, p0, n1, p1, float(np.float32(R.refpin_eval_mis(0, n0, p0, n1, p1))).hex()])
out["eval_mis_balance"] = mis
json.dump(out, open(os.path.join(os.path.dirname(__file__), "refpin_golden.json"), "w"), indent=0)
print("wrote refpin_golden.json:", {k: (len(v) if hasattr(v, '__len__') else v) for k, v in out.items() if k != 'source'})
