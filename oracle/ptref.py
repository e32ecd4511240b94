"""ORACLE binding (test infrastructure only): ctypes wrapper around oracle/_ref/libptref.so and librefpin.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. It is the checker,
never the thing measured or shipped.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_ref", "libptref.so")
_PIN = os.path.join(_HERE, "_ref", "librefpin.so")


def build(quiet=True):
    """Compile the oracle (and, when /root/reference exists, the reference pin) into oracle/_ref/."""
    r = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if not quiet:
        print(r.stdout)


def _load(path):
    if not os.path.exists(path):
        build()
    return ctypes.CDLL(path)


_lib = None
_lib16 = None
_LIB16 = os.path.join(_HERE, "_ref", "libptref_lp16.so")


def lib16():
    """The restatement of the reference's DEFAULT build: lp types in 16 bits (oracle/Makefile: -DPT_LP16=1). Same API as lib()."""
    global _lib16
    if _lib16 is None:
        L = _load(_LIB16)
        L.ptref_create.restype = ctypes.c_void_p
        L.ptref_radiance.restype = ctypes.POINTER(ctypes.c_float)
        L.ptref_num_tris.restype = ctypes.c_uint32
        _lib16 = L
    return _lib16


def lib():
    global _lib
    if _lib is None:
        L = _load(_LIB)
        L.ptref_create.restype = ctypes.c_void_p
        L.ptref_radiance.restype = ctypes.POINTER(ctypes.c_float)
        for n in ("ptref_hash32", "ptref_hash32_combine", "ptref_sobol", "ptref_owen_scramble", "ptref_f32tof16", "ptref_num_tris"):
            getattr(L, n).restype = ctypes.c_uint32
        L.ptref_hash32_to_float.restype = ctypes.c_float
        L.ptref_f16tof32.restype = ctypes.c_float
        L.ptref_f32tof16.argtypes = [ctypes.c_float]
        _lib = L
    return _lib


def refpin():
    """The reference's own compiled sources (None when unavailable, e.g. on the GPU box without a prebuilt .so)."""
    if not os.path.exists(_PIN):
        if os.path.isdir("/root/reference/Rtxpt/Shaders"):
            build()
        if not os.path.exists(_PIN):
            return None
    L = ctypes.CDLL(_PIN)
    for n in ("refpin_hash32", "refpin_hash32_combine", "refpin_sobol"):
        getattr(L, n).restype = ctypes.c_uint32
    L.refpin_hash32_to_float.restype = ctypes.c_float
    L.refpin_eval_mis.restype = ctypes.c_float
    L.refpin_eval_mis.argtypes = [ctypes.c_int] + [ctypes.c_float] * 4
    return L


_PIN_HLSL = os.path.join(os.path.dirname(_PIN), "librefpin_hlsl.so")


def refpin_hlsl():
    """Functions of the reference's .hlsli files compiled verbatim (oracle/refpin/hlsl_tu.py); None when unavailable."""
    if not os.path.exists(_PIN_HLSL):
        if os.path.isdir("/root/reference/Rtxpt/Shaders"):
            build()
        if not os.path.exists(_PIN_HLSL):
            return None
    return ctypes.CDLL(_PIN_HLSL)


def pin_call(fn, inputs, reference=False):
    """Evaluates pinned function `fn` (oracle/refpin/pin_fns.h) on rows of float32 inputs: the oracle's restatement, or the reference text itself."""
    L = refpin_hlsl() if reference else lib()
    if L is None:
        return None
    nin, nout = PIN_ARITY[fn]
    a = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, nin)
    out = np.zeros((a.shape[0], nout), np.float32)
    f = L.refhlsl_call if reference else L.ptref_pin_call
    f(ctypes.c_int(fn), a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(a.shape[0]), out.ctypes.data_as(ctypes.c_void_p))
    return out


def bsdf_probe(rows, reference=False):
    """Whole-BSDF probe over rows of tests/pin_inputs.bsdf_cases: the oracle's StandardBSDF, or the reference's BxDF.hlsli text (FalcorBSDF)."""
    L = refpin_hlsl() if reference else lib()
    if L is None:
        return None
    f = L.refhlsl_bsdf_probe if reference else L.ptref_bsdf_probe
    rows = np.ascontiguousarray(rows, np.float32)
    out = np.zeros((rows.shape[0], 10), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for i in range(rows.shape[0]):
        r = rows[i]
        f(vp(r[:14].copy()), int(r[14]), int(r[15]), vp(r[16:19].copy()), vp(r[19:22].copy()), int(r[22]), vp(out[i]))
    return out


def sample_streams(cases, reference=False):
    """cases: uint32 rows [packedPixel, vertexIndex, sampleIndex, effectSeed, kind, n<=8] -> float32 [len(cases), 8] (unused slots 0)."""
    L = refpin_hlsl() if reference else lib()
    if L is None:
        return None
    f = L.refhlsl_sample_stream if reference else L.ptref_sample_stream
    out = np.zeros((len(cases), 8), np.float32)
    for i, c in enumerate(cases):
        f(ctypes.c_uint32(int(c[0])), ctypes.c_uint32(int(c[1])), ctypes.c_uint32(int(c[2])), ctypes.c_uint32(int(c[3])), ctypes.c_int(int(c[4])), ctypes.c_uint32(int(c[5])),
          out[i].ctypes.data_as(ctypes.c_void_p))
    return out


LIGHT_PROBE_IO = {0: (3, 5), 1: (12, 12), 2: (17, 12), 3: (18, 1), 4: (3, 4)}      # words in / out per kind (oracle/refpin/hlsl_wrappers.inc)


def light_probe(kind, words, reference=False):
    """Polymorphic-light probe over rows of 32-bit words (floats as their bit patterns)."""
    L = refpin_hlsl() if reference else lib()
    if L is None:
        return None
    ni, no = LIGHT_PROBE_IO[kind]
    a = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, ni)
    out = np.zeros((a.shape[0], no), np.uint32)
    f = L.refhlsl_light_probe if reference else L.ptref_light_probe
    f(ctypes.c_int(kind), a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(a.shape[0]), out.ctypes.data_as(ctypes.c_void_p))
    return out


_pin_pt = {}


def pt_variant(settings=None):
    """The shader-macro combination (Sample.cpp:988-1040) that corresponds to a PtSettings record: (DiffuseBrdf, RR, firefly filter, nested quality, LD sampler, NEE)."""
    if settings is None:
        return (2, 1, 1, 1, 1, 1)
    g = lambda k: int(np.asarray(settings[k]).reshape(-1)[0])
    return (g("diffuseBrdf"), 1 if g("enableRussianRoulette") else 0, 1 if float(np.asarray(settings["fireflyFilterThreshold"]).reshape(-1)[0]) != 0 else 0,
            g("nestedDielectricsQuality"), 1 if g("enableLDSamplerForBSDF") else 0, 1 if g("NEEEnabled") else 0)


def refpin_pt(variant=(2, 1, 1, 1, 1, 1), lp16=False, mode=0):
    """The reference's integrator text (PathTracer.hlsli & its include closure) compiled over the oracle's scene services, for one macro combination;
    exports the whole ptref_* API plus refpt_render. Built on demand from /root/reference (oracle/refpin/hlsl_tu.py --integrator); None when unavailable."""
    key = tuple(variant) + (bool(lp16), int(mode))      # mode: PATH_TRACER_MODE (0 reference, 1 stable-plane build pass, 2 fill pass)
    if key in _pin_pt:
        return _pin_pt[key]
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(_PIN), "librefpin_pt_%d%d%d%d%d%d%s%s.so" % (tuple(variant) + ("_lp16" if lp16 else "", "_m%d" % mode if mode else "")))
    srcs = [os.path.join(here, "refpin", f) for f in ("hlsl_tu.py", "hlsl_shim.h", "hlsl_pt_stubs.h", "hlsl_pt_bridge_stubs.h", "hlsl_pt_wrappers.inc", "hlsl_lbfb_stubs.h", "hlsl_envbake_stubs.h", "hlsl_emisb_stubs.h")] + [os.path.join(here, "ptref", f) for f in os.listdir(os.path.join(here, "ptref"))]
    stale = not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in srcs)
    if stale:
        if not os.path.isdir("/root/reference/Rtxpt/Shaders"):
            if not os.path.exists(path):
                _pin_pt[key] = None
                return None
        else:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            d = "-DDiffuseBrdf=%d -DPT_ENABLE_RUSSIAN_ROULETTE=%d -DRTXPT_FIREFLY_FILTER=%d -DRTXPT_NESTED_DIELECTRICS_QUALITY=%d -DRTXPT_ENABLE_LOW_DISCREPANCY_SAMPLER_FOR_BSDF=%d -DPT_NEE_ENABLED=%d" % tuple(variant) + (" -DRTXPT_LP_TYPES_USE_16BIT_PRECISION=1 -DPT_LP16=1" if lp16 else "")
            if mode: d += " -DPATH_TRACER_MODE=%d" % mode
            cmd = ("python3 %s/refpin/hlsl_tu.py --integrator /root/reference | g++ -O2 -std=c++17 -fPIC -shared -fopenmp -mfma -ffp-contract=off -fno-fast-math "
                   "-fsingle-precision-constant -fpermissive -w %s -I%s/refpin -x c++ - -o %s" % (here, d, here, path))
            r = subprocess.run(["bash", "-o", "pipefail", "-c", cmd], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("librefpin_pt build failed:\n" + r.stderr[-3000:])
    L = ctypes.CDLL(path)
    L.ptref_create.restype = ctypes.c_void_p
    L.ptref_radiance.restype = ctypes.POINTER(ctypes.c_float)
    L.ptref_num_tris.restype = ctypes.c_uint32
    _pin_pt[key] = L
    return L


def bc6_encode(texels, reference=False, quality=False):
    """BC6UCompress.hlsl's EncodeP1 (the cube compressor's "Fast" mode; quality=True: CSMain with QUALITY 1, EncodeP1 + the best two-region partition) on blocks of
    16 RGB texels, float32 [n, 16, 3] -> uint32 [n, 4]; reference=True: the reference's own text (librefpin_pt), otherwise the oracle's restatement."""
    t = np.ascontiguousarray(texels, np.float32).reshape(-1, 48); out = np.zeros((len(t), 4), np.uint32)
    if reference:
        L = refpin_pt()
        if L is None: return None
        (L.refpt_bc6_encode_quality if quality else L.refpt_bc6_encode)(t.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(t)), out.ctypes.data_as(ctypes.c_void_p))
    else:
        (lib().ptref_bc6_encode_quality if quality else lib().ptref_bc6_encode)(t.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(t)), out.ctypes.data_as(ctypes.c_void_p))
    return out


def bc6_decode(blocks):
    """BC6H_UF16 blocks of the modes the cube compressor writes (11; 7.6 and 9.5 with two regions), uint32 [n, 4] -> the half bit patterns a fetch returns, uint32 [n, 16, 3]."""
    b = np.ascontiguousarray(blocks, np.uint32).reshape(-1, 4); out = np.zeros((len(b), 16, 3), np.uint32)
    lib().ptref_bc6_decode(b.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(b)), out.ctypes.data_as(ctypes.c_void_p))
    return out


def surface_probe(oracle_ctx, prims, uv_dir_cone):
    """Bridge::loadSurface of the reference text (PathTracerBridgeDonut.hlsli:612-853 and what it calls) next to the oracle's loadSurface for the same hits.
    oracle_ctx: an Oracle(reference_integrator=True) with a scene; prims: global triangle ids; uv_dir_cone: rows [u, v, dir.xyz, coneWidth, coneSpread].
    Returns (reference, oracle) as uint32 [n, 45]."""
    prims = np.ascontiguousarray(prims, np.uint32); a = np.ascontiguousarray(uv_dir_cone, np.float32).reshape(-1, 7)
    R = np.zeros((len(prims), 45), np.uint32); Q = np.zeros((len(prims), 45), np.uint32)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    oracle_ctx.L.refpt_surface_probe(oracle_ctx.h, ctypes.c_uint32(len(prims)), vp(prims), vp(a), vp(R), vp(Q))
    return R, Q


def lightbake_probe(kind, words, pyramid=None, color_mul=(1.0, 1.0, 1.0), distant_vs_local=0.0002, reference=False):
    """Light-baker functions (oracle/refpin/hlsl_wrappers.inc refhlsl_lightbake_probe). kind 0: 12-word light records -> ComputeWeight bits;
    kind 1: rows [dim, x, y, lightIndex, depthLimit] over an importance pyramid (list of (d, d, 4) float32 arrays, finest first) -> 5 words."""
    L = refpin_hlsl() if reference else lib()
    if L is None:
        return None
    ni, no = (12, 1) if kind == 0 else (5, 5)
    a = np.ascontiguousarray(words, np.uint32).reshape(-1, ni); out = np.zeros((a.shape[0], no), np.uint32)
    pyr = [np.ascontiguousarray(m, np.float32) for m in (pyramid or [])]
    ptrs = (ctypes.c_void_p * max(1, len(pyr)))(*[m.ctypes.data for m in pyr]) if pyr else (ctypes.c_void_p * 1)()
    dims = np.array([m.shape[0] for m in pyr] or [0], np.uint32); cm = np.array(color_mul, np.float32)
    f = L.refhlsl_lightbake_probe if reference else L.ptref_lightbake_probe
    f(ctypes.c_int(kind), a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(a.shape[0]), out.ctypes.data_as(ctypes.c_void_p), ptrs, dims.ctypes.data_as(ctypes.c_void_p),
      ctypes.c_uint32(len(pyr)), cm.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(distant_vs_local))
    return out


_PIN_MAT = os.path.join(os.path.dirname(_PIN), "librefpin_mat.so")


def frustum_planes(view_proj, reference=False):
    """LightsBaker::UpdateFrustumConsts' five normalised clip planes of a row-vector view-projection matrix (float32 [5, 4]); reference=True: the reference's C++ text
    (librefpin_mat.so), otherwise the oracle's restatement (neeat.h light_frustum_planes_from_viewproj). None when the reference library is unavailable."""
    m = np.ascontiguousarray(view_proj, np.float32).reshape(16); out = np.zeros((5, 4), np.float32)
    if reference:
        if not os.path.exists(_PIN_MAT):
            if os.path.isdir("/root/reference/Rtxpt/Shaders"): build()
            if not os.path.exists(_PIN_MAT): return None
        ctypes.CDLL(_PIN_MAT).reflight_frustum_planes(_p(m), _p(out))
    else:
        lib().ptref_frustum_planes(_p(m), _p(out))
    return out


def importance_boost(lights12, planes, mul, fade, weights, hist=None, delta_mul=0.0, reference=False):
    """LightsBaker.hlsl ImportanceBooster on n lights (uint32 [n, 12]: PolymorphicLightInfo + Ex): frustum term with the given planes, then the intensity-delta term against
    last frame's weights `hist` (None: off). reference=True: the reference's text (librefpin_pt), otherwise the oracle's restatement."""
    l = np.ascontiguousarray(lights12, np.uint32).reshape(-1, 12); pl = np.ascontiguousarray(planes, np.float32).reshape(20); wt = np.ascontiguousarray(weights, np.float32)
    h = None if hist is None else np.ascontiguousarray(hist, np.float32); out = np.zeros(len(l), np.float32)
    L = refpin_pt() if reference else lib()
    if L is None: return None
    f = L.refpt_importance_boost if reference else L.ptref_importance_boost
    f.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]; f.restype = None
    f(len(l), _p(l), _p(pl), float(mul), float(fade), float(delta_mul), _p(h), _p(wt), _p(out))
    return out


def proxy_counts(weights, usage=None, total_max_feedback=0, global_feedback_weight=0.0, sampling_type=1, reference=False):
    """LightsBaker.hlsl ComputeProxyCounts: per-light proxy counts (uint32 [n]) and their total from the light weights; usage: n + 1 counts of last frame's feedback (NEE-AT) or
    None. reference=True: the reference's text dispatched in groups of 128 threads (librefpin_pt), otherwise the oracle's build_light_proxies. The weight sum is taken in light
    order on both sides (the reference accumulates it with a float atomic, i.e. in no particular order)."""
    w = np.ascontiguousarray(weights, np.float32); n = len(w); u = None if usage is None else np.ascontiguousarray(usage, np.uint32); counts = np.zeros(n, np.uint32); ws = ctypes.c_float(0)
    f = lib().ptref_proxy_counts; f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p]
    total = f(n, _p(w), ctypes.byref(ws), _p(u), int(total_max_feedback), float(global_feedback_weight), int(sampling_type), _p(counts))
    if not reference: return counts, int(total)
    L = refpin_pt()
    if L is None: return None
    g = L.refpt_proxy_counts; g.restype = ctypes.c_uint32
    g.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p]
    rc = np.zeros(n, np.uint32); rt = g(n, _p(w), ws, _p(u), int(total_max_feedback), float(global_feedback_weight), int(sampling_type), _p(rc))
    return rc, int(rt)


def reference_material_from_json(text, textures):
    """The reference's PTMaterial::Read + FillData (Rtxpt/Materials/MaterialsBaker.cpp, compiled as it stands: oracle/refpin/mat_stubs.h) on one
    `.material.json` document. textures: {file name: packed texture word} = what the texture cache could load. Returns (128 bytes PTMaterialData,
    [EnableAlphaTesting, ExcludeFromNEE, SkipRender, UseDonutEmissiveIntensity], 5 paths, 5 (sRGB, NormalMap) pairs) or None when unavailable."""
    if not os.path.exists(_PIN_MAT):
        if os.path.isdir("/root/reference/Rtxpt/Shaders"):
            build()
        if not os.path.exists(_PIN_MAT):
            return None
    L = ctypes.CDLL(_PIN_MAT)
    names = list(textures)
    cn = (ctypes.c_char_p * max(1, len(names)))(*[n.encode() for n in names]) if names else (ctypes.c_char_p * 1)()
    cw = (ctypes.c_uint32 * max(1, len(names)))(*[int(textures[n]) & 0xFFFFFFFF for n in names]) if names else (ctypes.c_uint32 * 1)()
    out = (ctypes.c_uint8 * 128)(); flags = (ctypes.c_uint32 * 4)(); paths = ctypes.create_string_buffer(5 * 256); sn = (ctypes.c_uint32 * 10)()
    r = L.refmat_from_json(text.encode(), cn, cw, ctypes.c_int(len(names)), out, flags, paths, sn)
    if r != 0:
        raise ValueError("reference parser rejected the document")
    return bytes(out), [int(f) for f in flags], [paths.raw[256 * i:256 * i + 256].split(b"\0")[0].decode() for i in range(5)], [(int(sn[2 * i]), int(sn[2 * i + 1])) for i in range(5)]


def reference_convert_light(desc_words):
    """LightsBaker::ConvertLight (Rtxpt/Lighting/LightsBaker.cpp:456-556 and its helpers, compiled as they stand) on one light. desc_words: the 15 words of the
    product's PtAnalyticLightDesc. Returns (8 words, 4 words), "assert" for what the reference asserts on (spot with radius 0), or None when unavailable."""
    if not os.path.exists(_PIN_MAT):
        if os.path.isdir("/root/reference/Rtxpt/Shaders"):
            build()
        if not os.path.exists(_PIN_MAT):
            return None
    L = ctypes.CDLL(_PIN_MAT)
    d = np.ascontiguousarray(desc_words, np.uint32); base = np.zeros(8, np.uint32); ex = np.zeros(4, np.uint32)
    r = L.reflight_convert(d.ctypes.data_as(ctypes.c_void_p), base.ctypes.data_as(ctypes.c_void_p), ex.ctypes.data_as(ctypes.c_void_p))
    return "assert" if r == 2 else (base, ex)


def reference_color_transform(white_balance, white_point, auto_exposure, exposure_compensation, film_speed, shutter, f_number):
    """ToneMappingPass::UpdateWhiteBalanceTransform + UpdateColorTransform (Rtxpt/ToneMapper/ToneMappingPasses.cpp:392-441) over ColorUtils.h, compiled as they
    stand on Donut math stand-ins (oracle/refpin/color_stubs.inc); returns the nine floats in constant-buffer order (:344-347), or None when unavailable."""
    if not os.path.exists(_PIN_MAT):
        if os.path.isdir("/root/reference/Rtxpt/Shaders"):
            build()
        if not os.path.exists(_PIN_MAT):
            return None
    L = ctypes.CDLL(_PIN_MAT)
    if not hasattr(L, "refcolor_transform"):
        return None
    out = np.zeros(9, np.float32)
    L.refcolor_transform.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int] + [ctypes.c_float] * 4 + [ctypes.c_void_p]
    L.refcolor_transform(int(white_balance), white_point, int(auto_exposure), exposure_compensation, film_speed, shutter, f_number, out.ctypes.data_as(ctypes.c_void_p))
    return out


def reference_tonemap_constants(ui_words, avg_luminance, enabled):
    """ToneMappingPass::PreRender (SetParameters, UpdateExposureValue, UpdateWhiteBalanceTransform, UpdateColorTransform) and the constant fill of ::Render
    (Rtxpt/ToneMapper/ToneMappingPasses.cpp:186-193, 316-348, 373-441), compiled as they stand, on the 15 words of the product's PtToneMappingParameters.
    Returns the first 21 words of ToneMappingConstants (8 scalars, 3x4 colour transform, enabled), or None when unavailable."""
    if not os.path.exists(_PIN_MAT):
        if os.path.isdir("/root/reference/Rtxpt/Shaders"):
            build()
        if not os.path.exists(_PIN_MAT):
            return None
    L = ctypes.CDLL(_PIN_MAT)
    if not hasattr(L, "reftonemap_constants"):
        return None
    u = np.ascontiguousarray(ui_words, np.uint32); out = np.zeros(21, np.uint32)
    L.reftonemap_constants.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    L.reftonemap_constants(u.ctypes.data_as(ctypes.c_void_p), avg_luminance, int(enabled), out.ctypes.data_as(ctypes.c_void_p))
    return out


def reference_scene_leaves(text):
    """The leaves of a `.scene.json` graph through the reference's own ExtendedSceneTypeFactory::CreateLeaf and Load functions (ExtendedScene.cpp), then the
    scene-dependent part of Sample::SceneLoaded / UpdateCameraFromScene (Sample.cpp:457-479, 547-549, 566-573, 613-629), compiled as they stand over Donut
    stand-ins (oracle/refpin/scene_stubs.inc). Returns a dict, or None when unavailable."""
    if not os.path.exists(_PIN_MAT):
        if os.path.isdir("/root/reference/Rtxpt/Shaders"):
            build()
        if not os.path.exists(_PIN_MAT):
            return None
    L = ctypes.CDLL(_PIN_MAT)
    if not hasattr(L, "refscene_leaves"):
        return None
    I = np.zeros(16, np.int32); F = np.zeros(24, np.float32); lights = np.zeros((16, 12), np.uint32); path = ctypes.create_string_buffer(260)
    L.refscene_leaves.argtypes = [ctypes.c_char_p] + [ctypes.c_void_p] * 4
    r = L.refscene_leaves(text.encode(), I.ctypes.data_as(ctypes.c_void_p), F.ctypes.data_as(ctypes.c_void_p), lights.ctypes.data_as(ctypes.c_void_p), path)
    if r != 0:
        raise RuntimeError("refscene_leaves: %d" % r)
    return {"numLights": int(I[0]), "numCameras": int(I[1]), "hasEnvironment": int(I[2]), "envTextureIndex": int(I[3]), "selectedCameraIndex": int(I[4]), "realtimeMode": int(I[5]),
            "enableAnimations": int(I[6]), "realtimeFireflyFilterEnabled": int(I[7]), "bounceCount": int(I[8]), "diffuseBounceCount": int(I[9]), "autoExposure": int(I[10]),
            "proxies": int(I[11]), "directional": int(I[12]), "envRadianceScale": F[0:3].copy(), "envRotation": float(F[3]), "envPath": path.value.decode(),
            "cameraPos": F[4:7].copy(), "cameraTarget": F[7:10].copy(), "cameraUp": F[10:13].copy(), "verticalFov": float(F[13]), "zNear": float(F[14]),
            "exposureCompensation": float(F[15]), "exposureValue": float(F[16]), "exposureValueMin": float(F[17]), "exposureValueMax": float(F[18]),
            "realtimeFireflyFilterThreshold": float(F[19]), "texLODBias": float(F[20]), "lights": lights[:min(int(I[0]), 16)].copy()}


def reference_tonemap_defaults():
    """ToneMappingParameters{} of the reference (ToneMappingPasses.h:36-53) as the 15 words of PtToneMappingParameters, or None when unavailable."""
    if not os.path.exists(_PIN_MAT):
        return None
    L = ctypes.CDLL(_PIN_MAT)
    if not hasattr(L, "reftonemap_defaults"):
        return None
    out = np.zeros(15, np.uint32); L.reftonemap_defaults(out.ctypes.data_as(ctypes.c_void_p)); return out


def average_luminance(rgba):
    """ORACLE (numpy float64) of the auto-exposure luminance capture: ToneMappingPasses.cpp:78-97 (target lowered to powers of two), luminance_ps.hlsl:10-26
    (log2(max(1e-4, dot(color, (0.299, 0.587, 0.114)))) of the colour target through the linear sampler, clamp addressing), mip chain of 2x2 averages down to 1x1
    (for power-of-two sides: the plain mean), capture_cs (ToneMapping.hlsl:25-34) and exp2 on the host (ToneMappingPasses.cpp:284). rgba: (H, W, 4)."""
    img = np.asarray(rgba, np.float64)[..., :3]
    H, W = img.shape[:2]
    LW, LH = 1 << int(np.floor(np.log2(W))), 1 << int(np.floor(np.log2(H)))
    sx = (np.arange(LW) + 0.5) / LW * W - 0.5; sy = (np.arange(LH) + 0.5) / LH * H - 0.5
    x0 = np.floor(sx); y0 = np.floor(sy); fx = (sx - x0)[None, :, None]; fy = (sy - y0)[:, None, None]
    xa = np.clip(x0.astype(int), 0, W - 1); xb = np.clip(x0.astype(int) + 1, 0, W - 1); ya = np.clip(y0.astype(int), 0, H - 1); yb = np.clip(y0.astype(int) + 1, 0, H - 1)
    top = img[ya][:, xa] * (1 - fx) + img[ya][:, xb] * fx; bot = img[yb][:, xa] * (1 - fx) + img[yb][:, xb] * fx
    col = top * (1 - fy) + bot * fy
    lum = col @ np.array([0.299, 0.587, 0.114])
    return float(2.0 ** np.mean(np.log2(np.maximum(1e-4, lum))))


def _pin_table():
    """(names, arities) parsed from oracle/refpin/pin_fns.h so that Python never holds a second copy of the table"""
    import re
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "refpin", "pin_fns.h")).read()
    names = re.findall(r"^\s*PIN_(\w+)(?:\s*=\s*0)?,", text, re.M)
    names = [n for n in names if n != "COUNT"]
    ar = re.search(r"kPinArity\[PIN_COUNT\]\[2\] = \{[^\n]*\n(.*?)\n\};", text, re.S).group(1)
    pairs = [(int(a), int(b)) for a, b in re.findall(r"\{(\d+),\s*(\d+)\}", ar)]
    assert len(names) == len(pairs), (len(names), len(pairs))
    return names, pairs


PIN_NAMES, PIN_ARITY = _pin_table()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Oracle:
    """Mirrors the call order of Sample::Render: set scene -> set camera/settings -> render(sample range) -> radiance."""

    def __init__(self, reference_integrator=False, settings=None, lp16=False, mode=0):
        """reference_integrator=True: the same oracle scene services, but the path between the hits runs the REFERENCE'S integrator text
        (oracle/_ref/librefpin_pt_*.so, oracle/refpin/hlsl_pt_wrappers.inc), compiled for the shader-macro combination `settings` stand for;
        only where that library can be built."""
        self.reference_integrator = reference_integrator
        self.lp16 = bool(lp16)
        self.L = refpin_pt(pt_variant(settings), lp16=lp16, mode=mode) if reference_integrator else (lib16() if lp16 else lib())      # lp16: the reference's default build, lp types in 16 bits
        if self.L is None:
            raise RuntimeError("librefpin_pt.so not available (needs /root/reference)")
        self.h = ctypes.c_void_p(self.L.ptref_create())
        self.w = self.h_ = 0

    def close(self):
        if self.h:
            self.L.ptref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, sc):
        L, h = self.L, self.h
        self._keep = sc
        L.ptref_clear_textures(h)
        for (w, hh, fmt, px) in sc["textures"]:
            L.ptref_add_texture(h, w, hh, fmt, _p(px))
        L.ptref_set_materials(h, _p(sc["materials"]), len(sc["materials"]))
        L.ptref_set_geometry(h, _p(sc["indices"]), sc["indices"].size, _p(sc["positions"]), _p(sc["uvs"]), _p(sc["normals"]), _p(sc["tangents"]),
                             sc["positions"].shape[0], _p(sc["geometries"]), len(sc["geometries"]), _p(sc["meshes"]), len(sc["meshes"]))
        L.ptref_set_instances(h, _p(sc["instances"]), len(sc["instances"]))
        if sc.get("env") is not None or sc.get("env_cube_source") is not None:
            rgb, tw, cm = sc["env"] if sc.get("env") is not None else sc["env_cube_source"]      # "env_cube_source": (faces float32 [6, d, d, 4], transform, colour multiplier)
            # EnvMapSceneParams::ColorMultiplier as Sample.cpp:1936-1948 fills it: tint * intensity / c_envMapRadianceScale (the cube holds radiance * 1/4)
            cm4 = (np.asarray(cm, np.float32) * np.float32(4.0)).astype(np.float32)
            if sc.get("env") is not None: L.ptref_set_environment(h, _p(rgb), rgb.shape[1], rgb.shape[0], _p(tw), _p(cm4))
            else:
                faces = np.ascontiguousarray(rgb, np.float32); assert faces.ndim == 4 and faces.shape[0] == 6 and faces.shape[1] == faces.shape[2] and faces.shape[3] == 4
                L.ptref_set_environment_cube(h, _p(faces), faces.shape[1], _p(tw), _p(cm4))
            dl = sc.get("env_directional_lights")
            dl = np.ascontiguousarray(dl, np.float32).reshape(-1, 8) if dl is not None else np.zeros((0, 8), np.float32)
            L.ptref_set_environment_bake(h, int(sc.get("env_cube_dim", 256)), _p(dl) if len(dl) else None, len(dl))
            L.ptref_set_environment_compression(h, int(sc.get("env_compression", 0)))
        else:
            L.ptref_set_environment(h, None, 0, 0, None, None)
        if sc.get("sky") is not None:
            self.set_procedural_sky(sc["sky"]["consts"], sc["sky"].get("textures"))
            if sc.get("env") is None and sc.get("env_cube_source") is None: L.ptref_set_environment_bake(h, int(sc.get("env_cube_dim", 256)), None, 0)
        if sc.get("lights") is not None:
            base, ex = sc["lights"]
            L.ptref_set_lights(h, _p(base), _p(ex), len(base))

    def set_procedural_sky(self, consts, textures=None):
        import ctypes
        if consts is None: self.L.ptref_set_procedural_sky(self.h, None, None, None); return
        cbuf = np.frombuffer(bytes(consts), np.float32).copy() if isinstance(consts, ctypes.Structure) else np.ascontiguousarray(consts, np.float32).reshape(40)
        if textures is None: self.L.ptref_set_procedural_sky(self.h, _p(cbuf), None, None); return
        arrs = [np.ascontiguousarray(a, np.float32) for a in textures]
        dims = np.array([[a.shape[-2], a.shape[-3], a.shape[0] if a.ndim == 4 else 1] for a in arrs], np.uint32)
        ptrs = (ctypes.c_void_p * 4)(*[a.ctypes.data for a in arrs])
        self.L.ptref_set_procedural_sky(self.h, _p(cbuf), ptrs, _p(dims))

    def set_local_light_sampling(self, table=None, jitter=(0, 0), ratio=0.65, ssc_threshold=0.3, feedback=False):
        """NEE-AT inputs; table: uint32 [tilesY, tilesX, 128] packed entries or None"""
        import ctypes
        f = self.L.ptref_set_local_light_sampling
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_uint32] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_int]; f.restype = None
        if table is None: f(self.h, None, 0, 0, 0, 0, float(ratio), float(ssc_threshold), 1 if feedback else 0); return
        t = np.ascontiguousarray(table, np.uint32); assert t.ndim == 3 and t.shape[2] == 128
        f(self.h, _p(t), t.shape[1], t.shape[0], int(jitter[0]), int(jitter[1]), float(ratio), float(ssc_threshold), 1 if feedback else 0)

    def light_feedback(self, sample=0):
        w = np.zeros((self.h_, self.w), np.float32); c = np.zeros((self.h_, self.w), np.uint32)
        if not self.L.ptref_get_light_feedback(self.h, int(sample), _p(w), _p(c)): raise RuntimeError("no feedback for that sample")
        return w, c

    def set_light_importance_boost(self, view_proj=None, mul=8.0, fade_distance=5.0):
        """ImportanceBooster's frustum term; view_proj: the host's 4 x 4 view-projection matrix (row vectors: clip = p @ M) or None (off)"""
        import ctypes
        f = self.L.ptref_set_light_importance_boost; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float]; f.restype = None
        m = None if view_proj is None else np.ascontiguousarray(view_proj, np.float32).reshape(16)
        f(self.h, _p(m), float(mul), float(fade_distance))

    def set_view_projection(self, world_to_clip=None):
        import ctypes
        f = self.L.ptref_set_view_projection; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; f.restype = None
        m = None if world_to_clip is None else np.ascontiguousarray(world_to_clip, np.float32).reshape(16)
        f(self.h, _p(m))

    def set_neeat(self, enable=True, global_feedback_weight=0.75, ratio=0.65, ssc_threshold=0.3, prefilter=True):
        """NEE-AT with the baker in the loop: every sample of render() is a frame (feedback passes, then the path tracer)"""
        import ctypes
        f = self.L.ptref_set_neeat; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int]; f.restype = None
        f(self.h, 1 if enable else 0, float(global_feedback_weight), float(ratio), float(ssc_threshold), 1 if prefilter else 0)

    def neeat_reset(self): self.L.ptref_neeat_reset(self.h)

    def neeat_feedback(self):
        """the run's reservoirs as they stand: (total weight float32 [h, w], candidates uint32 [h, w])"""
        w = np.zeros((self.h_, self.w), np.float32); c = np.zeros((self.h_, self.w), np.uint32)
        if not self.L.ptref_neeat_get_feedback(self.h, _p(w), _p(c)): raise RuntimeError("no NEE-AT frame yet")
        return w, c

    def neeat_update_begin(self):
        """realtime mode: LightsBaker::UpdateBegin, before the frame's build pass (with reference_integrator=True the reference's baker text runs)"""
        (self.L.refpt_neeat_update_begin if self.reference_integrator else self.L.ptref_neeat_update_begin)(self.h)

    def neeat_update_end(self, depth, motion_vectors):
        """realtime mode: LightsBaker::UpdateEnd on the build pass's depth [h, w] f32 and screen-space motion vectors [h, w, 4] binary16 bit patterns of THIS frame; the fill passes
        that follow sample the tiles it leaves and fill the run's reservoirs"""
        d = np.ascontiguousarray(depth, np.float32); m = np.ascontiguousarray(motion_vectors, np.uint16)
        assert d.shape == (self.h_, self.w) and m.shape == (self.h_, self.w, 4)
        (self.L.refpt_neeat_update_end if self.reference_integrator else self.L.ptref_neeat_update_end)(self.h, _p(d), _p(m))

    def neeat_tables(self):
        """(tile table uint32 [tilesY, tilesX, 128], jitter (x, y), global proxy counters) of the last frame"""
        txy = np.zeros(2, np.uint32); jxy = np.zeros(2, np.uint32)
        if not self.L.ptref_neeat_get_tables(self.h, _p(txy), _p(jxy), None, None): raise RuntimeError("no NEE-AT frame yet")
        t = np.zeros((int(txy[1]), int(txy[0]), 128), np.uint32); pc = np.zeros(len(self.lights()["lights"]), np.uint32)
        self.L.ptref_neeat_get_tables(self.h, None, None, _p(t), _p(pc))
        return t, (int(jxy[0]), int(jxy[1])), pc

    def sky_eval(self, mode, rows):
        """mode 0: ProceduralSkyLowRes (x, y, face, direction) -> (n, 4); 1: atmosphere + sun (.., direction) -> (n, 3); 2: GetSkyRadianceToPoint (.., point) -> (n, 6)"""
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 6); out = np.zeros((len(rows), (4, 3, 6)[mode]), np.float32)
        self.L.ptref_sky_eval(self.h, int(mode), len(rows), _p(rows), _p(out)); return out

    def set_instances(self, inst):
        self._inst = inst
        self.L.ptref_set_instances(self.h, _p(inst), len(inst))

    def set_previous_pose(self, instances=None, positions=None):
        """The previous frame's instance transforms / vertex positions (arrays shaped like the scene's; None = did not move): what the stable-plane build pass's motion vectors
        see as object motion (Bridge::loadSurface's prevPosW)."""
        self._prev = (None if instances is None else np.ascontiguousarray(instances), None if positions is None else np.ascontiguousarray(positions, np.float32))
        self.L.ptref_set_previous_pose(self.h, _p(self._prev[0]) if self._prev[0] is not None else None, 0 if self._prev[0] is None else len(self._prev[0]),
                                       _p(self._prev[1]) if self._prev[1] is not None else None, 0 if self._prev[1] is None else self._prev[1].shape[0])

    def set_camera(self, cam):
        self._cam = np.ascontiguousarray(cam)
        self.L.ptref_set_camera(self.h, _p(self._cam))

    def set_settings(self, s):
        want16 = bool(int(np.asarray(s["useFp16Types"]).reshape(-1)[0])) if "useFp16Types" in (np.asarray(s).dtype.names or ()) else False
        if not self.reference_integrator and want16 != self.lp16:
            raise ValueError("settings ask for useFp16Types=%d but this Oracle was created with lp16=%s (two libraries: Oracle(lp16=True) restates the 16-bit build)" % (want16, self.lp16))
        self._set = np.ascontiguousarray(s)
        self.L.ptref_set_settings(self.h, _p(self._set))

    def resize(self, w, h):
        self.w, self.h_ = w, h
        self.L.ptref_resize(self.h, w, h)

    def reset_accumulation(self):
        self.L.ptref_reset_accumulation(self.h)

    def set_brute_force(self, enable):
        """Diagnostics: every ray query tests every triangle (the definition any BVH has to reproduce)."""
        self.L.ptref_set_brute_force(self.h, int(enable))

    def render(self, first, n, rect=None):
        if self.reference_integrator:
            if rect is None: self.L.refpt_render(self.h, first, n)
            else: self.L.refpt_render_rect(self.h, first, n, *rect)
        elif rect is None:
            self.L.ptref_render(self.h, first, n)
        else:
            self.L.ptref_render_rect(self.h, first, n, *rect)

    def build_stable_planes(self, sample_index, params):
        """The realtime mode's pre-pass (PATH_TRACER_MODE_BUILD_STABLE_PLANES) over the whole frame: dict of header [4, h, w] u32, planes [3 * plane stride] records of 80 bytes
        (tiled-swizzled order), stable_radiance / motion_vectors [h, w, 4] binary16 bit patterns, depth / spec_hit_t [h, w] f32, throughput [h, w] R11G11B10.
        params: a record of rtxpt_amd.scenes.STABLE_PLANES_PARAMS_DTYPE. With reference_integrator=True the reference's own text of that pass runs."""
        w, h = self.w, self.h_
        self.L.ptref_stable_planes_plane_stride.restype = ctypes.c_uint32
        stride = int(self.L.ptref_stable_planes_plane_stride(w, h))
        prm = np.ascontiguousarray(params)
        out = dict(header=np.zeros((4, h, w), np.uint32), planes=np.zeros((3 * stride, 20), np.uint32), stable_radiance=np.zeros((h, w, 4), np.uint16), depth=np.zeros((h, w), np.float32),
                   spec_hit_t=np.zeros((h, w), np.float32), motion_vectors=np.zeros((h, w, 4), np.uint16), throughput=np.zeros((h, w), np.uint32))
        fn = self.L.refpt_build_stable_planes if self.reference_integrator else self.L.ptref_build_stable_planes
        fn(self.h, int(sample_index), _p(prm), _p(out["header"]), _p(out["planes"]), _p(out["stable_radiance"]), _p(out["depth"]), _p(out["spec_hit_t"]), _p(out["motion_vectors"]), _p(out["throughput"]))
        out["plane_stride"] = stride
        return out

    def fill_stable_planes(self, sample_index, params, frame):
        """One sub-sample of the realtime mode's noisy pass (PATH_TRACER_MODE_FILL_STABLE_PLANES) over `frame` (what build_stable_planes returned): the planes' noisy radiance
        (PackedNoisyRadianceAndSpecAvg) and spec_hit_t are updated in place. With reference_integrator=True (mode=2) the reference's own text of that pass runs."""
        prm = np.ascontiguousarray(params)
        fn = self.L.refpt_fill_stable_planes if self.reference_integrator else self.L.ptref_fill_stable_planes
        fn(self.h, int(sample_index), _p(prm), _p(frame["header"]), _p(frame["planes"]), _p(frame["spec_hit_t"]))
        return frame

    def radiance(self):
        p = self.L.ptref_radiance(self.h)
        return np.ctypeslib.as_array(p, shape=(self.h_, self.w, 4)).copy()

    def counters(self):
        c = (ctypes.c_uint64 * 7)()
        self.L.ptref_get_counters(self.h, c)
        k = ("extendRays", "shadowRays", "hits", "nodeVisitsExt", "triTestsExt", "nodeVisitsSh", "triTestsSh")
        return dict(zip(k, [int(x) for x in c]))

    def num_tris(self):
        return int(self.L.ptref_num_tris(self.h))

    def trace_closest(self, rays, brute=False):
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.zeros((rays.shape[0], 4), np.float32)
        self.L.ptref_trace_closest(self.h, _p(rays), rays.shape[0], _p(out), 1 if brute else 0)
        return out

    def trace_visibility(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.zeros(rays.shape[0], np.uint32)
        self.L.ptref_trace_visibility(self.h, _p(rays), rays.shape[0], _p(out))
        return out

    def lights(self):
        n, np_, dim = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self.L.ptref_get_lights(self.h, ctypes.byref(n), ctypes.byref(np_), None, None, None, None, None, ctypes.byref(dim))
        lights = np.zeros((n.value, 8), np.uint32)
        ex = np.zeros((n.value, 4), np.uint32)
        pc = np.zeros(n.value, np.uint32)
        pi = np.zeros(np_.value, np.uint32)
        el = np.zeros(dim.value * dim.value, np.uint32)
        self.L.ptref_get_lights(self.h, None, None, _p(lights), _p(ex), _p(pc), _p(pi), _p(el), None)
        return dict(lights=lights, lightsEx=ex, proxyCounters=pc, proxyIndices=pi, envLookup=el, envLookupDim=dim.value)

    def subinstances(self):
        n = ctypes.c_uint32()
        self.L.ptref_get_subinstances(self.h, ctypes.byref(n), None)
        out = np.zeros((n.value, 8), np.uint32)
        self.L.ptref_get_subinstances(self.h, None, _p(out))
        return out

    def env_cube(self, reference=False):
        """The RGBA16F environment cube as uint32 [texels, 2] plus (dim, mipLevels); reference=True: baked by the reference's EnvMapBaker.hlsl text
        (needs an Oracle(reference_integrator=True) library)."""
        dim, lv = ctypes.c_uint32(), ctypes.c_uint32()
        n = self.L.ptref_get_env_cube(self.h, None, 0, ctypes.byref(dim), ctypes.byref(lv))
        out = np.zeros((n, 2), np.uint32)
        if n:
            if reference: self.L.refpt_env_bake(self.h, _p(out), n)
            else: self.L.ptref_get_env_cube(self.h, _p(out), n, None, None)
        return out, dim.value, lv.value

    def env_importance(self, dim, reference=False):
        """Level 0 of the radiance / importance map (EnvMapImportanceSamplingBaker.hlsl BuildMIPDescentImportanceMapCS, RGBA16F store) as float32 [dim, dim, 4];
        reference=True: computed by the reference's shader text (needs an Oracle(reference_integrator=True) library)."""
        out = np.zeros((dim, dim, 4), np.float32)
        (self.L.refpt_env_importance if reference else self.L.ptref_get_env_importance)(self.h, ctypes.c_uint32(dim), _p(out))
        return out

    def env_eval(self, dirs_lod):
        """EnvMap::EvalLocal (cube fetch x ColorMultiplier) on rows (localDir.xyz, lod) -> float32 [n, 3]."""
        a = np.ascontiguousarray(dirs_lod, np.float32).reshape(-1, 4); out = np.zeros((a.shape[0], 3), np.float32)
        self.L.ptref_env_eval(self.h, _p(a), ctypes.c_uint32(a.shape[0]), _p(out))
        return out

    def surface_probe(self, prims, uv_dir_cone):
        """loadSurface of this library's build on given hits: uint32 [n, 45] (layout of refpt_surface_probe). prims: global triangle ids; rows [u, v, dir.xyz, coneWidth, coneSpread]."""
        prims = np.ascontiguousarray(prims, np.uint32); a = np.ascontiguousarray(uv_dir_cone, np.float32).reshape(-1, 7)
        out = np.zeros((len(prims), 45), np.uint32)
        self.L.ptref_surface_probe(self.h, ctypes.c_uint32(len(prims)), _p(prims), _p(a), _p(out))
        return out

    def camera_ray(self, px, py, sample_index):
        o = (ctypes.c_float * 6)()
        self.L.ptref_camera_ray(self.h, px, py, sample_index, o)
        return np.array(o, np.float32)


def stable_planes_merge(frame, reference=False):
    """PostProcess.hlsl's NO_DENOISER_FINAL_MERGE over a frame of build_stable_planes / fill_stable_planes: stable radiance + every plane's noisy radiance -> float32 [h, w, 4]"""
    hd = np.ascontiguousarray(frame["header"]); _, h, w = hd.shape; out = np.zeros((h, w, 4), np.float32)
    pl = np.ascontiguousarray(frame["planes"]); sr = np.ascontiguousarray(frame["stable_radiance"])
    (refpin_pt().refpt_stable_planes_merge if reference else lib().ptref_stable_planes_merge)(w, h, _p(hd), _p(pl), _p(sr), _p(out))
    return out


def denoise_spec_hit_t(depth, spec_hit_t, reference=False):
    """DenoisingGuidesBaker::DenoiseSpecHitT on whole planes (float32 [h, w]): returns the filled-in specular hit distances. reference=True: the reference's compute shader text (any pin library)."""
    d = np.ascontiguousarray(depth, np.float32); t = np.array(spec_hit_t, np.float32, copy=True, order="C"); h, w = d.shape
    if reference: refpin_pt().refpt_denoise_spec_hit_t(w, h, _p(d), _p(t))
    else: lib().ptref_denoise_spec_hit_t(w, h, _p(d), _p(t))
    return t


def num_threads():
    return int(lib().ptref_num_threads())


TONEMAP_DTYPE = np.dtype([("whiteScale", "<f4"), ("whiteMaxLuminance", "<f4"), ("toneMapOperator", "<u4"), ("clamped", "<u4"),
                          ("autoExposure", "<u4"), ("avgLuminance", "<f4"), ("autoExposureLumValueMin", "<f4"), ("autoExposureLumValueMax", "<f4"),
                          ("colorTransform", "<f4", (9,)), ("enabled", "<u4"), ("_pad0", "<u4"), ("_pad1", "<u4")])


def tonemap_linear(rgba, params, reference=False):
    """applyToneMapping before the SRGBA8 store, as floats: the oracle's tm_apply, or ToneMapping.ps.hlsli itself (librefpin_hlsl.so)."""
    L = refpin_hlsl() if reference else lib()
    if L is None:
        return None
    a = np.ascontiguousarray(rgba, dtype=np.float32).reshape(-1, 4)
    out = np.zeros_like(a)
    p = np.ascontiguousarray(params)
    f = L.refhlsl_tonemap if reference else L.ptref_tonemap_linear
    f(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(a.shape[0]), p.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def tonemap(rgba, params):
    """ptref_tonemap: (..., 4) float32 radiance -> (..., 4) uint8 sRGB through the restated ToneMapping.ps.hlsli."""
    L = lib()
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    n = a.size // 4
    out = np.empty(n, dtype=np.uint32)
    p = np.ascontiguousarray(params)
    L.ptref_tonemap.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    L.ptref_tonemap.restype = None
    L.ptref_tonemap(a.ctypes.data_as(ctypes.c_void_p), n, p.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out.view(np.uint8).reshape(a.shape[:-1] + (4,))

