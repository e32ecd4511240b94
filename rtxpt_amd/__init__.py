"""rtxpt_amd — host-side mirror of the reference's render-pass seam over the C-ABI of libmi355pt.so.

The product is the shared library (hand-written HIP for gfx950, built in-tree by `rtxpt_amd/csrc/Makefile`); this
package is the thin Python host binding used by the tests and bench.py: same call order as `Sample::Render`
(/root/reference/Rtxpt/Sample.cpp:1891-2313): load/set scene -> set camera/settings -> render(sample range) -> map radiance.

There is no CPU fallback: importing works anywhere, but creating a `PathTracer` without the built library or without
a HIP device raises.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import scenes  # noqa: F401  (scene generators + data-contract dtypes)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355PT_LIB", os.path.join(_HERE, "libmi355pt.so"))   # MI355PT_LIB: developer A/B builds only
HOOKS_LIB_PATH = os.path.join(_HERE, "libmi355pt_testhooks.so")                  # the tests' build: + include/mi355pt_testhooks.h
_libs = {}

PT_OK = 0
PT_ERROR_INVALID_ARGUMENT, PT_ERROR_NO_DEVICE, PT_ERROR_HIP, PT_ERROR_IO, PT_ERROR_UNSUPPORTED, PT_ERROR_NOT_READY = 1, 2, 3, 4, 5, 6
PT_TEX_RGBA8_UNORM, PT_TEX_RGBA8_SRGB, PT_TEX_RGBA32F = 0, 1, 2
STATUS = {0: "PT_OK", 1: "PT_ERROR_INVALID_ARGUMENT", 2: "PT_ERROR_NO_DEVICE", 3: "PT_ERROR_HIP", 4: "PT_ERROR_IO", 5: "PT_ERROR_UNSUPPORTED", 6: "PT_ERROR_NOT_READY"}

# every symbol include/mi355pt.h declares
EXPORTS = [
    "pt_create", "pt_destroy", "pt_get_last_error", "pt_load_scene_gltf", "pt_gltf_animation_load", "pt_gltf_animation_instances", "pt_gltf_animation_positions", "pt_gltf_animation_normals", "pt_animate_normals", "pt_gltf_animation_free", "pt_set_geometry", "pt_set_instances", "pt_set_materials",
    "pt_set_environment", "pt_set_environment_cube", "pt_image_read_dds_cube", "pt_set_environment_bake", "pt_set_environment_compression", "pt_set_procedural_sky", "pt_procedural_sky_default_params", "pt_procedural_sky_update", "pt_env_bake_lights", "pt_set_lights", "pt_set_light_importance_boost", "pt_set_local_light_sampling", "pt_get_light_feedback", "pt_set_neeat", "pt_set_view_projection", "pt_neeat_reset", "pt_get_neeat_tables", "pt_neeat_pack_feedback", "pt_neeat_unpack_feedback", "pt_bridge_camera", "pt_set_camera", "pt_default_settings", "pt_set_settings", "pt_animate",
    "pt_resize", "pt_render", "pt_reset_accumulation", "pt_map_radiance", "pt_unmap_radiance", "pt_shard_info", "pt_pack_shard",
    "pt_unpack_shard", "pt_device_radiance", "pt_trace_closest", "pt_trace_visibility", "pt_get_lights", "pt_get_env_cube", "pt_get_subinstances",
    "pt_get_scene_info", "pt_get_build_stats", "pt_get_bvh_info", "pt_set_counters",
    "pt_default_tonemap", "pt_tonemap", "pt_image_read_float", "pt_image_free", "pt_image_read_dds", "pt_image_read_dds_memory", "pt_image_read_jpeg", "pt_scene_import_texture", "pt_write_png", "pt_write_bmp", "pt_set_fused_traversal", "pt_set_serial_kernels", "pt_set_tail_paths", "pt_animate_ranges", "pt_set_motion_history", "pt_set_previous_pose", "pt_realtime_frame", "pt_exchange_planes_host", "pt_neeat_update_begin", "pt_neeat_update_end", "pt_pack_stable_plane_guides", "pt_unpack_stable_plane_guides", "pt_stable_planes_shard_bytes", "pt_pack_stable_planes", "pt_unpack_stable_planes", "pt_gather_stable_planes", "pt_material_from_json", "pt_convert_light",
    "pt_tonemap_color_transform", "pt_scene_json_import", "pt_scene_import_free", "pt_scene_import_cameras", "pt_scene_import_lights",
    "pt_scene_import_directional_lights", "pt_scene_import_instances", "pt_scene_import_geometries", "pt_scene_import_materials", "pt_scene_import_vertices", "pt_set_scene_directional_lights", "pt_scene_import_apply", "pt_scene_import_settings", "pt_average_luminance",
    "pt_default_tone_mapping_parameters", "pt_tonemap_from_parameters", "pt_scene_import_tone_mapping",
    "pt_stable_planes_plane_stride", "pt_build_stable_planes", "pt_fill_stable_planes", "pt_denoise_spec_hit_t", "pt_stable_planes_merge", "pt_get_stable_planes",
    "pt_comm_unique_id", "pt_comm_init", "pt_comm_destroy", "pt_gather", "pt_shard_layout", "pt_gather_host", "pt_neeat_exchange_host",
]
TEST_HOOK_EXPORTS = ["pt_probe"]      # include/mi355pt_testhooks.h: libmi355pt_testhooks.so only


class PtError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("%s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


class PtGeometryBuffers(ctypes.Structure):
    _fields_ = [("indices", ctypes.c_void_p), ("numIndices", ctypes.c_uint32), ("positions", ctypes.c_void_p), ("uvs", ctypes.c_void_p),
                ("normals", ctypes.c_void_p), ("tangents", ctypes.c_void_p), ("numVertices", ctypes.c_uint32)]


class PtTextureDesc(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("format", ctypes.c_uint32), ("pixels", ctypes.c_void_p)]


class PtEnvMapSceneParams(ctypes.Structure):
    _fields_ = [("Transform", ctypes.c_float * 12), ("ColorMultiplier", ctypes.c_float * 3), ("Enabled", ctypes.c_float)]


class PtProceduralSkyConstants(ctypes.Structure):      # SampleProceduralSky.hlsli:18-46 (40 floats; as_array() gives them as numpy)
    _fields_ = [("StarIrradiance", ctypes.c_float * 3), ("StarAngularDiameter", ctypes.c_float), ("RayleightScatteringRGB", ctypes.c_float * 3), ("PlanetSurfaceRadius", ctypes.c_float),
                ("MieScatteringRGB", ctypes.c_float * 3), ("PlanetAtmosphereRadius", ctypes.c_float), ("MieHenyeyGreensteinG", ctypes.c_float), ("SqDistanceToHorizontalBoundary", ctypes.c_float),
                ("AtmosphereHeight", ctypes.c_float), ("reserved", ctypes.c_float),
                ("FinalRadianceMultiplier", ctypes.c_float * 3), ("_padding3", ctypes.c_float), ("SunDir", ctypes.c_float * 3), ("CloudsTime", ctypes.c_float),
                ("GroundAlbedo", ctypes.c_float * 3), ("SunAngularDiameter", ctypes.c_float), ("_padding0", ctypes.c_float), ("_padding1", ctypes.c_float), ("sun_solid_angle", ctypes.c_float),
                ("_padding2", ctypes.c_float), ("physical_sky_ground_radiance", ctypes.c_float * 3), ("cloud_density_offset", ctypes.c_float),
                ("sky_transmittance", ctypes.c_float), ("sky_phase_g", ctypes.c_float), ("sky_amb_phase_g", ctypes.c_float), ("sky_scattering", ctypes.c_float)]

    def as_array(self): return np.frombuffer(bytes(self), np.float32).copy()


class PtSkyTexture(ctypes.Structure):
    _fields_ = [("rgba", ctypes.c_void_p), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("depth", ctypes.c_uint32), ("_pad", ctypes.c_uint32)]


class PtProceduralSkyTextures(ctypes.Structure):
    _fields_ = [("transmittance", PtSkyTexture), ("scattering", PtSkyTexture), ("irradiance", PtSkyTexture), ("clouds", PtSkyTexture)]


class PtProceduralSkyParams(ctypes.Structure):         # SampleProceduralSky.h:70-82
    _fields_ = [("colorTint", ctypes.c_float * 3), ("brightness", ctypes.c_float), ("sunBrightness", ctypes.c_float), ("cloudsMovementSpeed", ctypes.c_float), ("timeOfDayMovementSpeed", ctypes.c_float),
                ("sunTimeOfDayOffset", ctypes.c_float), ("sunEastWestRotation", ctypes.c_float), ("sunAngularDiameterDeg", ctypes.c_float), ("cloudDensityOffset", ctypes.c_float),
                ("cloudTransmittance", ctypes.c_float), ("cloudScattering", ctypes.c_float)]


class PtProceduralSkyState(ctypes.Structure):
    _fields_ = [("lastSceneTime", ctypes.c_double), ("timeOfDayL1", ctypes.c_float), ("timeOfDayL2", ctypes.c_float), ("lastConstants", PtProceduralSkyConstants)]


def _sky_textures(textures):
    """four float32 arrays, (h, w, 4) or (d, h, w, 4): transmittance, scattering, irradiance, clouds -> (PtProceduralSkyTextures, the arrays kept alive)"""
    keep, t = [], PtProceduralSkyTextures()
    for name, a in zip(("transmittance", "scattering", "irradiance", "clouds"), textures):
        a = np.ascontiguousarray(a, np.float32); assert a.shape[-1] == 4 and a.ndim in (3, 4)
        d = a.shape[0] if a.ndim == 4 else 1
        setattr(t, name, PtSkyTexture(a.ctypes.data, a.shape[-2], a.shape[-3], d, 0)); keep.append(a)
    return t, keep


def procedural_sky_default_params(lib=None):
    L = lib or load_library(); p = PtProceduralSkyParams(); L.pt_procedural_sky_default_params(ctypes.byref(p)); return p


def procedural_sky_update(state, scene_time, preset="==PROCEDURAL_SKY==", params=None, force_instant=False, lib=None):
    """SampleProceduralSky::Update: -> (PtProceduralSkyConstants, changed). `state`: a PtProceduralSkyState kept between calls (PtProceduralSkyState() to start)."""
    L = lib or load_library(); out = PtProceduralSkyConstants()
    L.pt_procedural_sky_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_char_p, ctypes.c_int32, ctypes.c_void_p]; L.pt_procedural_sky_update.restype = ctypes.c_int32
    r = L.pt_procedural_sky_update(ctypes.byref(state), ctypes.byref(params) if params is not None else None, float(scene_time), preset.encode() if preset is not None else None, 1 if force_instant else 0, ctypes.byref(out))
    if r < 0: raise RuntimeError("pt_procedural_sky_update failed: %s" % STATUS.get(-r, r))
    return out, bool(r)


class PtDeviceDesc(ctypes.Structure):
    _fields_ = [("deviceOrdinal", ctypes.c_int32), ("shardRank", ctypes.c_uint32), ("shardCount", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


class PtFrameStats(ctypes.Structure):
    _fields_ = [("extendRays", ctypes.c_uint64), ("shadowRays", ctypes.c_uint64), ("hits", ctypes.c_uint64),
                ("nodeVisitsExtend", ctypes.c_uint64), ("triTestsExtend", ctypes.c_uint64), ("nodeVisitsShadow", ctypes.c_uint64), ("triTestsShadow", ctypes.c_uint64),
                ("leafVisitsExtend", ctypes.c_uint64), ("waveItersExtend", ctypes.c_uint64), ("leafVisitsShadow", ctypes.c_uint64), ("waveItersShadow", ctypes.c_uint64),
                ("extendPhaseCycles", ctypes.c_uint64 * 4), ("leafBlocksExtend", ctypes.c_uint64), ("waveItersMaxExtend", ctypes.c_uint64), ("extendRayIterHist", ctypes.c_uint64 * 16), ("longRayCount", ctypes.c_uint32), ("_padLong", ctypes.c_uint32), ("longRays", (ctypes.c_float * 8) * 32), ("extendEvents", ctypes.c_uint64 * 8),
                ("gpuMilliseconds", ctypes.c_double), ("extendKernelMs", ctypes.c_double), ("shadeKernelMs", ctypes.c_double), ("shadowKernelMs", ctypes.c_double),
                ("extendLaunches", ctypes.c_uint32), ("iterations", ctypes.c_uint32), ("pathsTraced", ctypes.c_uint32), ("tailLaunches", ctypes.c_uint32)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if isinstance(getattr(self, n), ctypes.Array) else getattr(self, n)) for n, _ in self._fields_ if n != "_pad"}


def build_library(verbose=False):
    """Compile libmi355pt.so for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("libmi355pt.so build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH




def _share_hip_runtime_with_torch():
    """One HIP runtime per process. PyTorch-ROCm ships its own libamdhip64.so; were libmi355pt.so loaded first it would bind /opt/rocm's copy, a later
    `import torch` would bring a second runtime into the process, and whichever of the two initialises second no longer finds the device (seen on the
    MI355X box: load_library(); import torch; torch.cuda.is_available(); pt_create -> PT_ERROR_NO_DEVICE). Loading torch's copy first — without importing
    torch — makes both resolve to it by SONAME, in either import order. Without torch installed, /opt/rocm's copy is used as linked."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def _share_rccl_with_torch():
    """One RCCL per process, for the same reason as the HIP runtime: libmi355pt.so binds RCCL at run time (dlopen, RTLD_NOLOAD first), so loading torch's
    copy before pt_comm_unique_id / pt_comm_init makes the library and torch.distributed use the same one whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


COMM_ID_BYTES = 128


def comm_unique_id():
    """pt_comm_unique_id: 128 bytes created on rank 0; hand them to the other ranks (torch.distributed broadcast, MPI, a file) and call PathTracer.comm_init everywhere."""
    _share_rccl_with_torch()
    L = load_library()
    buf = (ctypes.c_ubyte * COMM_ID_BYTES)()
    r = L.pt_comm_unique_id(buf)
    if r != 0:
        raise PtError(r, "pt_comm_unique_id")
    return bytes(buf)


def shard_layout(width, height, rank, world):
    """pt_shard_layout: packed pixel ids (x<<16|y) owned by `rank`, in pack order (host only)."""
    L = load_library()
    n = ctypes.c_uint32()
    r = L.pt_shard_layout(width, height, rank, world, None, 0, ctypes.byref(n))
    if r != 0:
        raise PtError(r, "pt_shard_layout")
    out = np.zeros(n.value, np.uint32)
    r = L.pt_shard_layout(width, height, rank, world, _p(out), n.value, None)
    if r != 0:
        raise PtError(r, "pt_shard_layout")
    return out


class PtTransport(ctypes.Structure):
    SEND = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32)
    RECV = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32)
    GROUP = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p)
    _fields_ = [("user", ctypes.c_void_p), ("send", SEND), ("recv", RECV), ("group_begin", GROUP), ("group_end", GROUP)]


def gather_host(width, height, rank, world, rgba, send, recv):
    """pt_gather_host: the library's gather protocol over host memory and the caller's point-to-point transport.
    send(ptr, nbytes, peer) / recv(ptr, nbytes, peer) are Python callables taking raw addresses; rgba: (h, w, 4) float32, modified in place on rank 0."""
    L = load_library()
    assert rgba.dtype == np.float32 and rgba.flags["C_CONTIGUOUS"] and rgba.shape == (height, width, 4)

    def _wrap(fn):
        def cb(user, buf, nbytes, peer):
            try:
                fn(buf, nbytes, peer); return 0
            except Exception:      # an exception must not unwind through the C frames
                import traceback; traceback.print_exc(); return 1
        return cb
    t = PtTransport(None, PtTransport.SEND(_wrap(send)), PtTransport.RECV(_wrap(recv)), PtTransport.GROUP(), PtTransport.GROUP())
    r = L.pt_gather_host(width, height, rank, world, _p(rgba), ctypes.byref(t))
    if r != 0:
        raise PtError(r, "pt_gather_host")
    return rgba


def neeat_exchange_host(width, height, rank, world, total_weight, candidates, depth, send, recv):
    """pt_neeat_exchange_host: all ranks end up with all ranks' feedback reservoirs and exported depth; (h, w) float32 / uint32 / float32 planes, modified in place; send / recv as in gather_host"""
    L = load_library()
    assert total_weight.dtype == np.float32 and candidates.dtype == np.uint32 and depth.dtype == np.float32 and total_weight.shape == candidates.shape == depth.shape == (height, width)

    def _wrap(fn):
        def cb(user, buf, nbytes, peer):
            try:
                fn(buf, nbytes, peer); return 0
            except Exception:
                import traceback; traceback.print_exc(); return 1
        return cb
    t = PtTransport(None, PtTransport.SEND(_wrap(send)), PtTransport.RECV(_wrap(recv)), PtTransport.GROUP(), PtTransport.GROUP())
    r = L.pt_neeat_exchange_host(width, height, rank, world, _p(total_weight), _p(candidates), _p(depth), ctypes.byref(t))
    if r != 0:
        raise PtError(r, "pt_neeat_exchange_host")


def exchange_planes_host(width, height, rank, world, planes, send, recv, to_root=False):
    """pt_exchange_planes_host: `planes` = C-contiguous arrays of shape (height, width[, ...]) — any bytes per pixel; all-to-all (every rank ends up with every rank's pixels) or,
    with to_root, towards rank 0 only; modified in place; send / recv as in gather_host"""
    L = load_library()
    for a in planes: assert a.flags["C_CONTIGUOUS"] and a.shape[0] == height and a.shape[1] == width
    bpp = np.array([a.nbytes // (width * height) for a in planes], np.uint32)
    ptrs = (ctypes.c_void_p * len(planes))(*[a.ctypes.data for a in planes])

    def _wrap(fn):
        def cb(user, buf, nbytes, peer):
            try:
                fn(buf, nbytes, peer); return 0
            except Exception:
                import traceback; traceback.print_exc(); return 1
        return cb
    t = PtTransport(None, PtTransport.SEND(_wrap(send)), PtTransport.RECV(_wrap(recv)), PtTransport.GROUP(), PtTransport.GROUP())
    f = L.pt_exchange_planes_host; f.argtypes = [ctypes.c_uint32] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p]; f.restype = ctypes.c_int32
    r = f(width, height, rank, world, ptrs, _p(bpp), len(planes), 1 if to_root else 0, ctypes.byref(t))
    if r != 0:
        raise PtError(r, "pt_exchange_planes_host")


def kernel_source_digest():
    """SHA-256 over the sources of the library (rtxpt_amd/csrc/*.h, *.hip and *.cpp, names and contents, in name order): what ties a rocprofv3 counter summary under
    profiles/ to the code a bench.py run executes (tools/profile_round.sh writes it, bench.py compares it before quoting `bound` / `traffic`). The binary itself is tied by
    library_digest()."""
    import glob, hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def library_digest(test_hooks=False):
    """SHA-256 of the shared library file load_library() opens — the binary the numbers of a run come from (bench.py prints it, tools/profile_round.sh records it)."""
    import hashlib
    h = hashlib.sha256()
    with open(HOOKS_LIB_PATH if test_hooks else LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""): h.update(blk)
    return h.hexdigest()


def load_library(test_hooks=False):
    """dlopen libmi355pt.so (test_hooks: libmi355pt_testhooks.so, the tests' build with include/mi355pt_testhooks.h's evaluation hooks). Raises if it has not been built:
    the product path never falls back to anything else."""
    if _libs.get(test_hooks) is None:
        path = HOOKS_LIB_PATH if test_hooks else LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError("%s is missing (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C rtxpt_amd/csrc`." % (os.path.basename(path), path))
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(path)
        L.pt_get_last_error.restype = ctypes.c_char_p
        L.pt_get_last_error.argtypes = [ctypes.c_void_p]
        _libs[test_hooks] = L
    return _libs[test_hooks]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


TONEMAP_DTYPE = np.dtype([("whiteScale", "<f4"), ("whiteMaxLuminance", "<f4"), ("toneMapOperator", "<u4"), ("clamped", "<u4"),
                          ("autoExposure", "<u4"), ("avgLuminance", "<f4"), ("autoExposureLumValueMin", "<f4"), ("autoExposureLumValueMax", "<f4"),
                          ("colorTransform", "<f4", (9,)), ("enabled", "<u4"), ("_pad0", "<u4"), ("_pad1", "<u4")])
assert TONEMAP_DTYPE.itemsize == 80
TONEMAP_OPERATORS = {"linear": 0, "reinhard": 1, "reinhard_modified": 2, "heji_hable_alu": 3, "hable_uc2": 4, "aces": 5}


def default_tonemap(exposure_compensation=0.0, film_speed=100.0, shutter=1.0, f_number=1.0, **kw):
    """pt_default_tonemap: ToneMappingParameters defaults + manual-exposure colour transform (ToneMappingPasses.h:36-53, .cpp:428-441). No device needed."""
    L = load_library()
    t = np.zeros((), dtype=TONEMAP_DTYPE)
    L.pt_default_tonemap.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float]
    r = L.pt_default_tonemap(_p(t), exposure_compensation, film_speed, shutter, f_number)
    if r != 0:
        raise PtError(r, "pt_default_tonemap")
    for k, v in kw.items():
        t[k] = TONEMAP_OPERATORS[v] if (k == "toneMapOperator" and isinstance(v, str)) else v
    return t


TONE_MAPPING_PARAMETERS_DTYPE = np.dtype([("exposureMode", "<u4"), ("toneMapOperator", "<u4"), ("autoExposure", "<u4"), ("exposureCompensation", "<f4"), ("exposureValue", "<f4"),
                                          ("filmSpeed", "<f4"), ("fNumber", "<f4"), ("shutter", "<f4"), ("whiteBalance", "<u4"), ("whitePoint", "<f4"), ("whiteMaxLuminance", "<f4"),
                                          ("whiteScale", "<f4"), ("clamped", "<u4"), ("exposureValueMin", "<f4"), ("exposureValueMax", "<f4")])
assert TONE_MAPPING_PARAMETERS_DTYPE.itemsize == 60


def default_tone_mapping_parameters(**kw):
    """pt_default_tone_mapping_parameters: ToneMappingParameters{} (ToneMappingPasses.h:36-53) as a TONE_MAPPING_PARAMETERS_DTYPE record. No device needed."""
    L = load_library()
    u = np.zeros((), dtype=TONE_MAPPING_PARAMETERS_DTYPE)
    L.pt_default_tone_mapping_parameters.argtypes = [ctypes.c_void_p]
    r = L.pt_default_tone_mapping_parameters(_p(u))
    if r != 0:
        raise PtError(r, "pt_default_tone_mapping_parameters")
    for k, v in kw.items():
        u[k] = TONEMAP_OPERATORS[v] if (k == "toneMapOperator" and isinstance(v, str)) else v
    return u


def tonemap_from_parameters(ui, avg_luminance=1.0, enabled=True):
    """pt_tonemap_from_parameters: ToneMappingPass::PreRender + the constant fill of ::Render on a ToneMappingParameters block -> TONEMAP_DTYPE record."""
    L = load_library()
    t = np.zeros((), dtype=TONEMAP_DTYPE)
    L.pt_tonemap_from_parameters.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p]
    r = L.pt_tonemap_from_parameters(_p(ui), avg_luminance, 1 if enabled else 0, _p(t))
    if r != 0:
        raise PtError(r, "pt_tonemap_from_parameters")
    return t


def tonemap_color_transform(params, white_balance=False, white_point=6500.0, exposure_compensation=0.0, film_speed=100.0, shutter=1.0, f_number=1.0):
    """pt_tonemap_color_transform: UpdateWhiteBalanceTransform + UpdateColorTransform (ToneMappingPasses.cpp:392-441, ColorUtils.h:128-197) written into
    params["colorTransform"] (a TONEMAP_DTYPE record, modified in place and returned). No device needed."""
    L = load_library()
    L.pt_tonemap_color_transform.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_float] * 5
    r = L.pt_tonemap_color_transform(_p(params), 1 if white_balance else 0, white_point, exposure_compensation, film_speed, shutter, f_number)
    if r != 0:
        raise PtError(r, "pt_tonemap_color_transform")
    return params


class PtMaterialJsonInfo(ctypes.Structure):
    _fields_ = [("texturePath", (ctypes.c_char * 256) * 5), ("textureSRGB", ctypes.c_uint32 * 5), ("textureNormalMap", ctypes.c_uint32 * 5),
                ("enableAlphaTesting", ctypes.c_uint32), ("excludeFromNEE", ctypes.c_uint32), ("skipRender", ctypes.c_uint32), ("useDonutEmissiveIntensity", ctypes.c_uint32)]


def material_from_json(text, texture_words=(0xFFFFFFFF,) * 5):
    """pt_material_from_json: an RTXPT `.material.json` document -> (PTMaterialData record as numpy scalar, info dict). No device needed.
    texture_words: packed texture words of the Base / ORM / Normal / Emissive / Transmission textures (0xFFFFFFFF = not loaded)."""
    from . import scenes
    L = load_library()
    out = np.zeros((), dtype=scenes.MATERIAL_DTYPE)
    info = PtMaterialJsonInfo()
    words = (ctypes.c_uint32 * 5)(*[int(w) & 0xFFFFFFFF for w in texture_words])
    L.pt_material_from_json.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    r = L.pt_material_from_json(text.encode() if isinstance(text, str) else text, words, _p(out), ctypes.byref(info))
    if r != 0:
        raise PtError(r, "pt_material_from_json")
    return out, {"texturePath": [bytes(info.texturePath[i]).split(b"\0")[0].decode() for i in range(5)], "textureSRGB": list(info.textureSRGB),
                 "textureNormalMap": list(info.textureNormalMap), "enableAlphaTesting": bool(info.enableAlphaTesting), "excludeFromNEE": bool(info.excludeFromNEE),
                 "skipRender": bool(info.skipRender), "useDonutEmissiveIntensity": bool(info.useDonutEmissiveIntensity)}


class PtAnalyticLightDesc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint32), ("position", ctypes.c_float * 3), ("direction", ctypes.c_float * 3), ("color", ctypes.c_float * 3), ("intensity", ctypes.c_float),
                ("radius", ctypes.c_float), ("innerAngle", ctypes.c_float), ("outerAngle", ctypes.c_float)]


class GltfAnimation:
    """pt_gltf_animation_*: the animations of a .gltf / .glb file, evaluated on the host into the instance transforms pt_animate takes."""
    def __init__(self, path):
        self.L = load_library(); self.h = ctypes.c_void_p(); n = ctypes.c_uint32(); d = ctypes.c_float()
        self.L.pt_gltf_animation_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_float)]
        self.L.pt_gltf_animation_instances.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint32]
        self.L.pt_gltf_animation_free.argtypes = [ctypes.c_void_p]; self.L.pt_gltf_animation_free.restype = None
        r = self.L.pt_gltf_animation_load(str(path).encode(), ctypes.byref(self.h), ctypes.byref(n), ctypes.byref(d))
        if r != PT_OK: raise PtError(r, "pt_gltf_animation_load(%s)" % path)
        self.count, self.duration = n.value, d.value

    def instances(self, t, animation=0):
        """INSTANCE_DTYPE array of the scene's instances at time t [s]"""
        from . import scenes
        n = self.L.pt_gltf_animation_instances(self.h, animation, float(t), None, 0)
        if n < 0: raise PtError(-n, "pt_gltf_animation_instances")
        out = np.zeros(n, scenes.INSTANCE_DTYPE)
        if n: assert self.L.pt_gltf_animation_instances(self.h, animation, float(t), _p(out), n) == n
        return out

    def positions(self, t, animation=0):
        """pt_gltf_animation_positions: float32 [vertices, 3], the file's vertex stream with every skinned primitive posed at time t [s] (pt_animate's `positions`)."""
        self.L.pt_gltf_animation_positions.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint32]
        n = self.L.pt_gltf_animation_positions(self.h, animation, float(t), None, 0)
        if n < 0: raise PtError(-n, "pt_gltf_animation_positions")
        out = np.zeros((n, 3), np.float32)
        if n: assert self.L.pt_gltf_animation_positions(self.h, animation, float(t), _p(out), n) == n
        return out

    def normals(self, t, animation=0):
        """pt_gltf_animation_normals: (normals, tangents) uint32 [vertices], the posed NORMAL / TANGENT streams in pt_set_geometry's SNORM8 packing (pt_animate_normals' arguments)."""
        f = self.L.pt_gltf_animation_normals; f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        n = f(self.h, animation, float(t), None, None, 0)
        if n < 0: raise PtError(-n, "pt_gltf_animation_normals")
        nrm, tan = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        if n: assert f(self.h, animation, float(t), _p(nrm), _p(tan), n) == n
        return nrm, tan

    def close(self):
        if self.h: self.L.pt_gltf_animation_free(self.h); self.h = ctypes.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass


def read_dds(path):
    """pt_image_read_dds: the top mip level of a 2D .dds file. Returns (pixels, format): uint8 [h, w, 4] with PT_TEX_RGBA8_UNORM / PT_TEX_RGBA8_SRGB, or
    float32 [h, w, 4] with PT_TEX_RGBA32F. No device needed."""
    L = load_library()
    w, h, fmt = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(); p = ctypes.c_void_p()
    L.pt_image_read_dds.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_void_p)]
    L.pt_image_free.argtypes = [ctypes.c_void_p]; L.pt_image_free.restype = None
    r = L.pt_image_read_dds(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(fmt), ctypes.byref(p))
    if r != PT_OK: raise PtError(r, "pt_image_read_dds(%s)" % path)
    try:
        ct, dt = (ctypes.c_float, np.float32) if fmt.value == 2 else (ctypes.c_uint8, np.uint8)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), shape=(h.value, w.value, 4)).astype(dt, copy=True), int(fmt.value)
    finally:
        L.pt_image_free(p)


def read_dds_cube(path):
    """pt_image_read_dds_cube: the six faces (top level) of a float / BC6H cube-map .dds as float32 [6, dim, dim, 4] in D3D's face order — an environment source for
    the scene key "env_cube_source" (pt_set_environment_cube). No device needed."""
    L = load_library()
    dim = ctypes.c_uint32(); p = ctypes.c_void_p()
    L.pt_image_read_dds_cube.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_void_p)]
    L.pt_image_free.argtypes = [ctypes.c_void_p]; L.pt_image_free.restype = None
    r = L.pt_image_read_dds_cube(str(path).encode(), ctypes.byref(dim), ctypes.byref(p))
    if r != PT_OK: raise PtError(r, "pt_image_read_dds_cube(%s)" % path)
    try:
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), shape=(6, dim.value, dim.value, 4)).astype(np.float32, copy=True)
    finally:
        L.pt_image_free(p)


def read_jpeg(data):
    """pt_image_read_jpeg: a JPEG stream (bytes, or a path) as uint8 [h, w, 4]. No device needed."""
    if not isinstance(data, (bytes, bytearray)): data = open(data, "rb").read()
    L = load_library(); w, h = ctypes.c_uint32(), ctypes.c_uint32(); p = ctypes.c_void_p()
    L.pt_image_read_jpeg.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_void_p)]
    L.pt_image_free.argtypes = [ctypes.c_void_p]; L.pt_image_free.restype = None
    r = L.pt_image_read_jpeg(bytes(data), len(data), ctypes.byref(w), ctypes.byref(h), ctypes.byref(p))
    if r != PT_OK: raise PtError(r, "pt_image_read_jpeg")
    try:
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        L.pt_image_free(p)


def read_float_image(path):
    """pt_image_read_float: an OpenEXR (scan-line; none / RLE / ZIPS / ZIP) or Radiance .hdr file as float32 [h, w, 3], top row first. No device needed."""
    L = load_library()
    w, h = ctypes.c_uint32(), ctypes.c_uint32(); p = ctypes.POINTER(ctypes.c_float)()
    L.pt_image_read_float.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.POINTER(ctypes.c_float))]
    L.pt_image_free.argtypes = [ctypes.c_void_p]; L.pt_image_free.restype = None
    r = L.pt_image_read_float(str(path).encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(p))
    if r != PT_OK: raise PtError(r, "pt_image_read_float(%s)" % path)
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        L.pt_image_free(p)


def env_bake_lights(world_lights, cube_dim, transform=None):
    """pt_env_bake_lights (Sample::UpdateLighting): world-space rows of EMB_DirectionalLight -> the rows pt_set_environment_bake takes (angular size raised to
    what the cube resolves, direction in the environment's local frame). transform: the 12 floats of EnvMapSceneParams.Transform or None. No device needed."""
    L = load_library()
    a = np.ascontiguousarray(world_lights, np.float32).reshape(-1, 8); out = np.zeros_like(a)
    p = None
    if transform is not None:
        p = PtEnvMapSceneParams((ctypes.c_float * 12)(*np.asarray(transform, np.float32).tolist()), (ctypes.c_float * 3)(1.0, 1.0, 1.0), 1.0)
    r = L.pt_env_bake_lights(_p(a) if len(a) else None, len(a), ctypes.byref(p) if p is not None else None, int(cube_dim), _p(out) if len(a) else None)
    if r != PT_OK: raise PtError(r, "pt_env_bake_lights")
    return out


def convert_light(kind, position, color, intensity, radius, direction=(0.0, -1.0, 0.0), inner_angle=0.0, outer_angle=0.0):
    """pt_convert_light (LightsBaker::ConvertLight): kind "point" / "spot" -> (8 words PolymorphicLightInfo, 4 words PolymorphicLightInfoEx) as uint32 arrays. No device needed."""
    L = load_library()
    d = PtAnalyticLightDesc(); d.type = {"point": 0, "spot": 1}[kind]
    d.position[:] = [float(x) for x in position]; d.direction[:] = [float(x) for x in direction]; d.color[:] = [float(x) for x in color]
    d.intensity, d.radius, d.innerAngle, d.outerAngle = float(intensity), float(radius), float(inner_angle), float(outer_angle)
    base = np.zeros(8, np.uint32); ex = np.zeros(4, np.uint32)
    L.pt_convert_light.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    r = L.pt_convert_light(ctypes.byref(d), _p(base), _p(ex))
    if r != 0:
        raise PtError(r, "pt_convert_light")
    return base, ex


SCENE_CAMERA_DTYPE = np.dtype([("position", "<f4", 3), ("direction", "<f4", 3), ("up", "<f4", 3), ("verticalFov", "<f4"), ("zNear", "<f4"), ("exposureMask", "<u4"),
                               ("enableAutoExposure", "<u4"), ("exposureCompensation", "<f4"), ("exposureValue", "<f4"), ("exposureValueMin", "<f4"),
                               ("exposureValueMax", "<f4"), ("name", "S64")])
SCENE_JSON_INFO_DTYPE = np.dtype([("numModels", "<u4"), ("numGeometries", "<u4"), ("numMeshes", "<u4"), ("numInstances", "<u4"), ("numMaterials", "<u4"), ("numTextures", "<u4"),
                                  ("numLights", "<u4"), ("numCameras", "<u4"), ("materialOverrides", "<u4"), ("texturesNotLoaded", "<u4"), ("lightsDropped", "<u4"),
                                  ("lightProxies", "<u4"), ("skippedGeometries", "<u4"), ("directionalLights", "<u4"), ("hasEnvironment", "<u4"),
                                  ("envRadianceScale", "<f4", 3), ("envRotation", "<f4"), ("envTextureIndex", "<i4"), ("envPath", "S260"), ("settingsMask", "<u4"),
                                  ("realtimeMode", "<u4"), ("enableAnimations", "<u4"), ("startingCamera", "<i4"), ("realtimeFireflyFilter", "<f4"), ("maxBounces", "<i4"),
                                  ("maxDiffuseBounces", "<i4"), ("textureMIPBias", "<f4"), ("selectedCamera", "<i4"), ("lightProxiesResolved", "<u4")])
assert SCENE_CAMERA_DTYPE.itemsize == 132 and SCENE_JSON_INFO_DTYPE.itemsize == 380


class SceneImport:
    """pt_scene_json_import: an RTXPT `.scene.json` asset folder read on the host (ExtendedScene + Sample::SceneLoaded + MaterialsBaker::Load). No device
    needed; PathTracer.apply_scene_import hands it to a context."""

    def __init__(self, scene_path, media_path=None):
        from . import scenes
        self.L = load_library()
        self.h = ctypes.c_void_p()
        self.info = np.zeros((), dtype=SCENE_JSON_INFO_DTYPE)
        self.L.pt_scene_json_import.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
        r = self.L.pt_scene_json_import(str(scene_path).encode(), None if media_path is None else str(media_path).encode(), ctypes.byref(self.h), _p(self.info))
        if r != 0:
            raise PtError(r, "pt_scene_json_import")
        self.L.pt_scene_import_free.argtypes = [ctypes.c_void_p]; self.L.pt_scene_import_free.restype = None

        def fetch(fn, dtype, n):
            a = np.zeros(max(int(n), 1), dtype=dtype)
            f = getattr(self.L, fn); f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
            got = f(self.h, _p(a), int(n))
            assert got == n, (fn, got, n)
            return a[:n]
        I = self.info
        self.cameras = fetch("pt_scene_import_cameras", SCENE_CAMERA_DTYPE, I["numCameras"])
        self.instances = fetch("pt_scene_import_instances", scenes.INSTANCE_DTYPE, I["numInstances"])
        self.geometries = fetch("pt_scene_import_geometries", scenes.GEOMETRY_DTYPE, I["numGeometries"])
        self.materials = fetch("pt_scene_import_materials", scenes.MATERIAL_DTYPE, I["numMaterials"])
        self.L.pt_scene_import_vertices.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        nv = self.L.pt_scene_import_vertices(self.h, None, 0); self.positions = np.zeros((max(nv, 1), 3), np.float32)
        assert self.L.pt_scene_import_vertices(self.h, _p(self.positions), nv) == nv
        self.positions = self.positions[:nv]
        n = int(I["numLights"]); self.lights = np.zeros((max(n, 1), 8), np.uint32); self.lights_ex = np.zeros((max(n, 1), 4), np.uint32)
        self.L.pt_scene_import_lights.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        assert self.L.pt_scene_import_lights(self.h, _p(self.lights), _p(self.lights_ex), n) == n
        self.lights, self.lights_ex = self.lights[:n], self.lights_ex[:n]
        self.L.pt_scene_import_directional_lights.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        nd = self.L.pt_scene_import_directional_lights(self.h, None, 0)
        self.directional_lights = np.zeros((max(nd, 1), 8), np.float32)      # rows of EMB_DirectionalLight in world space: colour rgb, irradiance, direction xyz, angular size [rad]
        assert self.L.pt_scene_import_directional_lights(self.h, _p(self.directional_lights), nd) == nd
        self.directional_lights = self.directional_lights[:nd]

    def texture(self, index):
        """pt_scene_import_texture: the decoded top level of imported texture `index` as (uint8 [h, w, 4], format)."""
        d = PtTextureDesc()
        self.L.pt_scene_import_texture.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        r = self.L.pt_scene_import_texture(self.h, int(index), ctypes.byref(d))
        if r != 0: raise PtError(r, "pt_scene_import_texture(%d)" % index)
        return np.ctypeslib.as_array(ctypes.cast(d.pixels, ctypes.POINTER(ctypes.c_uint8)), shape=(d.height, d.width, 4)).copy(), int(d.format)

    def tone_mapping(self, ui=None, camera=-1):
        """pt_scene_import_tone_mapping: Sample::SceneLoaded's exposure defaults + Sample::UpdateCameraFromScene on a ToneMappingParameters record."""
        ui = default_tone_mapping_parameters() if ui is None else ui
        self.L.pt_scene_import_tone_mapping.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        r = self.L.pt_scene_import_tone_mapping(self.h, int(camera), _p(ui))
        if r != 0:
            raise PtError(r, "pt_scene_import_tone_mapping")
        return ui

    def apply_settings(self, settings):
        """pt_scene_import_settings: the scene's SampleSettings keys written over a SETTINGS_DTYPE record (in place)."""
        self.L.pt_scene_import_settings.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        r = self.L.pt_scene_import_settings(self.h, _p(settings))
        if r != 0:
            raise PtError(r, "pt_scene_import_settings")
        return settings

    def close(self):
        if self.h:
            self.L.pt_scene_import_free(self.h); self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_image(path, rgba8):
    """pt_write_png / pt_write_bmp by extension (the reference's screenshot formats). rgba8: (H, W, 4) uint8."""
    L = load_library()
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    fn = L.pt_write_bmp if path.lower().endswith(".bmp") else L.pt_write_png
    fn.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    r = fn(path.encode(), _p(a), a.shape[1], a.shape[0])
    if r != 0:
        raise PtError(r, "write_image " + path)


def bridge_camera(width, height, pos, direction, up, fov_y, near_z=0.01, far_z=1e5, focal_distance=10.0, aperture_radius=0.0, jitter=(0.0, 0.0)):
    """pt_bridge_camera: the library's BridgeCamera (PathTracerShared.h:109-141). Does not need a device."""
    L = load_library()
    cam = np.zeros((), dtype=scenes.CAMERA_DTYPE)
    f3 = lambda v: (ctypes.c_float * 3)(*[float(x) for x in v])
    j = (ctypes.c_float * 2)(float(jitter[0]), float(jitter[1]))
    r = L.pt_bridge_camera(ctypes.c_uint32(width), ctypes.c_uint32(height), f3(pos), f3(direction), f3(up), ctypes.c_float(fov_y), ctypes.c_float(near_z),
                           ctypes.c_float(far_z), ctypes.c_float(focal_distance), ctypes.c_float(aperture_radius), j, _p(cam))
    if r != PT_OK:
        raise PtError(r, "pt_bridge_camera")
    return cam


class PathTracer:
    """One pt_context (one GPU). Method names follow the C-ABI; the call order follows Sample::Render."""

    def __init__(self, device=0, shard_rank=0, shard_count=1, serial_kernels=False, prefer_fast_build=False, host_sah_builder=False, test_hooks=False):
        """serial_kernels: PT_DEVICE_SERIAL_KERNELS — pt_render runs one batch on one stream (clean per-kernel timings) instead of two pipelined half-frame batches.
        prefer_fast_build: PT_DEVICE_PREFER_FAST_BUILD — scene builds use the plain device-side PLOC builder instead of the default fast-trace tree (PLOC + parallel
        re-insertion + cost-driven wide nodes, on the device); host_sah_builder: PT_DEVICE_HOST_SAH_BUILDER — round 2's binned SAH + re-insertion on the host's cores.
        test_hooks: run on libmi355pt_testhooks.so, the tests' build of the same sources that also exports pt_probe (include/mi355pt_testhooks.h)."""
        self.L = load_library(test_hooks)
        self.test_hooks = test_hooks
        self.h = ctypes.c_void_p()
        desc = PtDeviceDesc(device, shard_rank, shard_count, (1 if serial_kernels else 0) | (2 if prefer_fast_build else 0) | (4 if host_sah_builder else 0))
        r = self.L.pt_create(ctypes.byref(desc), ctypes.byref(self.h))
        if r != PT_OK:
            raise PtError(r, "pt_create failed (no HIP device? this library has no CPU path)")
        self.width = self.height = 0
        self.shard_rank, self.shard_count = shard_rank, shard_count
        self._keep = []

    def _chk(self, r, what):
        if r != PT_OK:
            raise PtError(r, what + ": " + (self.L.pt_get_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            self.L.pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene
    def load_scene_gltf(self, path):
        self._chk(self.L.pt_load_scene_gltf(self.h, path.encode()), "pt_load_scene_gltf")

    def apply_scene_import(self, imp):
        """pt_scene_import_apply: materials, geometry, instances and analytic lights of a SceneImport (camera / environment / settings stay with the caller)."""
        self.L.pt_scene_import_apply.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self.L.pt_scene_import_apply(self.h, imp.h), "pt_scene_import_apply")

    def set_scene(self, sc):
        """sc: dict from rtxpt_amd.scenes (same arrays the oracle receives)."""
        self._keep = [sc]
        tex = (PtTextureDesc * max(1, len(sc["textures"])))()
        for i, (w, h, fmt, px) in enumerate(sc["textures"]):
            tex[i] = PtTextureDesc(w, h, fmt, px.ctypes.data)
        self._chk(self.L.pt_set_materials(self.h, _p(sc["materials"]), len(sc["materials"]), tex, len(sc["textures"])), "pt_set_materials")
        gb = PtGeometryBuffers(sc["indices"].ctypes.data, sc["indices"].size, sc["positions"].ctypes.data, sc["uvs"].ctypes.data,
                               sc["normals"].ctypes.data, sc["tangents"].ctypes.data, sc["positions"].shape[0])
        self._chk(self.L.pt_set_geometry(self.h, ctypes.byref(gb), _p(sc["geometries"]), len(sc["geometries"]), _p(sc["meshes"]), len(sc["meshes"])), "pt_set_geometry")
        self._chk(self.L.pt_set_instances(self.h, _p(sc["instances"]), len(sc["instances"])), "pt_set_instances")
        if sc.get("env") is not None or sc.get("env_cube_source") is not None:
            rgb, tw, cm = sc["env"] if sc.get("env") is not None else sc["env_cube_source"]      # "env_cube_source": (faces float32 [6, d, d, 4], transform, colour multiplier) — a cube map as the image
            # EnvMapSceneParams::ColorMultiplier as Sample.cpp:1936-1948 fills it: tint * intensity / c_envMapRadianceScale (the cube holds radiance * 1/4)
            cm4 = (np.asarray(cm, np.float32) * np.float32(4.0)).astype(np.float32)
            p = PtEnvMapSceneParams((ctypes.c_float * 12)(*tw.tolist()), (ctypes.c_float * 3)(*cm4.tolist()), 1.0)
            if sc.get("env") is not None:
                self._chk(self.L.pt_set_environment(self.h, _p(rgb), rgb.shape[1], rgb.shape[0], ctypes.byref(p)), "pt_set_environment")
            else:
                faces = np.ascontiguousarray(rgb, np.float32); assert faces.ndim == 4 and faces.shape[0] == 6 and faces.shape[1] == faces.shape[2] and faces.shape[3] == 4
                self._chk(self.L.pt_set_environment_cube(self.h, _p(faces), faces.shape[1], ctypes.byref(p)), "pt_set_environment_cube")
            dl = sc.get("env_directional_lights")       # rows of EMB_DirectionalLight: colour rgb, intensity, direction xyz, angular size
            dl = np.ascontiguousarray(dl, np.float32).reshape(-1, 8) if dl is not None else np.zeros((0, 8), np.float32)
            self._chk(self.L.pt_set_environment_bake(self.h, int(sc.get("env_cube_dim", 256)), _p(dl) if len(dl) else None, len(dl)), "pt_set_environment_bake")
            self._chk(self.L.pt_set_environment_compression(self.h, int(sc.get("env_compression", 0))), "pt_set_environment_compression")
        else:
            self._chk(self.L.pt_set_environment(self.h, None, 0, 0, None), "pt_set_environment")
        if sc.get("sky") is not None:                   # {"consts": PtProceduralSkyConstants or 40 floats, "textures": four arrays}: the procedural sky as the cube's source
            self.set_procedural_sky(sc["sky"]["consts"], sc["sky"].get("textures"))
            if sc.get("env") is None and sc.get("env_cube_source") is None: self._chk(self.L.pt_set_environment_bake(self.h, int(sc.get("env_cube_dim", 256)), None, 0), "pt_set_environment_bake")
        if sc.get("lights") is not None:
            base, ex = sc["lights"]
            self._chk(self.L.pt_set_lights(self.h, _p(base), _p(ex), len(base)), "pt_set_lights")

    def set_procedural_sky(self, consts, textures=None):
        """consts: PtProceduralSkyConstants / 40 float32 values / None (off); textures: four float32 arrays (transmittance, scattering, irradiance, clouds) or None (keep)"""
        if consts is None: self._chk(self.L.pt_set_procedural_sky(self.h, None, None), "pt_set_procedural_sky"); return
        cbuf = np.frombuffer(bytes(consts), np.float32).copy() if isinstance(consts, ctypes.Structure) else np.ascontiguousarray(consts, np.float32).reshape(40)
        t = keep = None
        if textures is not None: t, keep = _sky_textures(textures)
        self._chk(self.L.pt_set_procedural_sky(self.h, _p(cbuf), ctypes.byref(t) if t is not None else None), "pt_set_procedural_sky")

    def set_local_light_sampling(self, table=None, jitter=(0, 0), ratio=0.65, ssc_threshold=0.3, feedback=False):
        """NEE-AT inputs (pt_set_local_light_sampling). table: uint32 [tilesY, tilesX, 128] packed entries (light << 9 | count - 1, sorted per tile) or None (no local layer)"""
        f = self.L.pt_set_local_light_sampling
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_uint32] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_int32]; f.restype = ctypes.c_int32
        if table is None: self._chk(f(self.h, None, 0, 0, 0, 0, float(ratio), float(ssc_threshold), 1 if feedback else 0), "pt_set_local_light_sampling"); return
        t = np.ascontiguousarray(table, np.uint32)
        if t.ndim != 3 or t.shape[2] != 128: raise ValueError("local sampling table: uint32 [tilesY, tilesX, 128]")
        self._chk(f(self.h, _p(t), t.shape[1], t.shape[0], int(jitter[0]), int(jitter[1]), float(ratio), float(ssc_threshold), 1 if feedback else 0), "pt_set_local_light_sampling")

    def set_light_importance_boost(self, view_proj=None, mul=8.0, fade_distance=5.0):
        """pt_set_light_importance_boost: the frustum term of LightsBaker's ImportanceBooster; view_proj: the host's 4 x 4 view-projection matrix (row vectors: clip = p @ M) or None (off)"""
        f = self.L.pt_set_light_importance_boost; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float]; f.restype = ctypes.c_int32
        m = None if view_proj is None else np.ascontiguousarray(view_proj, np.float32).reshape(16)
        self._chk(f(self.h, _p(m), float(mul), float(fade_distance)), "pt_set_light_importance_boost")

    def set_view_projection(self, world_to_clip=None):
        """pt_set_view_projection: the host's world-to-clip matrix (row vectors), for the depth the reference-mode path tracer exports and NEE-AT's disocclusion test reads"""
        f = self.L.pt_set_view_projection; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; f.restype = ctypes.c_int32
        m = None if world_to_clip is None else np.ascontiguousarray(world_to_clip, np.float32).reshape(16)
        self._chk(f(self.h, _p(m)), "pt_set_view_projection")

    def build_stable_planes(self, sample_index, params):
        """pt_build_stable_planes + pt_get_stable_planes: the realtime mode's pre-pass over the frame. params: a record of scenes.STABLE_PLANES_PARAMS_DTYPE. Returns a dict of
        header [4, h, w] u32, planes [3 * plane stride, 20] u32 (records of scenes.STABLE_PLANE_DTYPE, tiled-swizzled order), stable_radiance / motion_vectors [h, w, 4] binary16 bit
        patterns, depth / spec_hit_t [h, w] f32, throughput [h, w] R11G11B10, plane_stride, stats."""
        w, h = self.width, self.height
        f = self.L.pt_stable_planes_plane_stride; f.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]; f.restype = ctypes.c_int32
        stride = ctypes.c_uint32(0); self._chk(f(w, h, ctypes.byref(stride)), "pt_stable_planes_plane_stride"); stride = int(stride.value)
        prm = np.ascontiguousarray(params); assert prm.dtype.itemsize == 224
        st = PtFrameStats()
        f = self.L.pt_build_stable_planes; f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]; f.restype = ctypes.c_int32
        self._chk(f(self.h, int(sample_index), _p(prm), ctypes.byref(st)), "pt_build_stable_planes")
        out = dict(header=np.zeros((4, h, w), np.uint32), planes=np.zeros((3 * stride, 20), np.uint32), stable_radiance=np.zeros((h, w, 4), np.uint16), depth=np.zeros((h, w), np.float32),
                   spec_hit_t=np.zeros((h, w), np.float32), motion_vectors=np.zeros((h, w, 4), np.uint16), throughput=np.zeros((h, w), np.uint32))
        f = self.L.pt_get_stable_planes; f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t] + [ctypes.c_void_p] * 5; f.restype = ctypes.c_int32
        self._chk(f(self.h, _p(out["header"]), _p(out["planes"]), 3 * stride, _p(out["stable_radiance"]), _p(out["depth"]), _p(out["spec_hit_t"]), _p(out["motion_vectors"]), _p(out["throughput"])), "pt_get_stable_planes")
        out["plane_stride"] = stride
        out["stats"] = st.as_dict()
        return out

    def fill_stable_planes(self, sample_index, params, sub_samples=1):
        """pt_fill_stable_planes for sub_samples consecutive sample indices + pt_get_stable_planes: the realtime mode's noisy passes over the frame build_stable_planes left.
        Returns the same dict as build_stable_planes (planes with their noisy radiance, spec_hit_t filled), stats summed over the sub-samples."""
        prm = np.ascontiguousarray(params); assert prm.dtype.itemsize == 224
        f = self.L.pt_fill_stable_planes; f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]; f.restype = ctypes.c_int32
        total = {}
        for s in range(sub_samples):
            st = PtFrameStats()
            self._chk(f(self.h, int(sample_index) + s, _p(prm), ctypes.byref(st)), "pt_fill_stable_planes")
            for k, v in st.as_dict().items():
                if np.isscalar(v): total[k] = total.get(k, 0) + v
        w, h = self.width, self.height
        g = self.L.pt_stable_planes_plane_stride; g.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]; g.restype = ctypes.c_int32
        stride = ctypes.c_uint32(0); self._chk(g(w, h, ctypes.byref(stride)), "pt_stable_planes_plane_stride"); stride = int(stride.value)
        out = dict(header=np.zeros((4, h, w), np.uint32), planes=np.zeros((3 * stride, 20), np.uint32), stable_radiance=np.zeros((h, w, 4), np.uint16), depth=np.zeros((h, w), np.float32),
                   spec_hit_t=np.zeros((h, w), np.float32), motion_vectors=np.zeros((h, w, 4), np.uint16), throughput=np.zeros((h, w), np.uint32))
        g = self.L.pt_get_stable_planes; g.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t] + [ctypes.c_void_p] * 5; g.restype = ctypes.c_int32
        self._chk(g(self.h, _p(out["header"]), _p(out["planes"]), 3 * stride, _p(out["stable_radiance"]), _p(out["depth"]), _p(out["spec_hit_t"]), _p(out["motion_vectors"]), _p(out["throughput"])), "pt_get_stable_planes")
        out["plane_stride"] = stride; out["stats"] = total
        return out

    def get_stable_planes(self):
        """pt_get_stable_planes: the buffers of the last build / fill / realtime pass (the dict build_stable_planes returns, without stats)"""
        w, h = self.width, self.height
        g = self.L.pt_stable_planes_plane_stride; g.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]; g.restype = ctypes.c_int32
        stride = ctypes.c_uint32(0); self._chk(g(w, h, ctypes.byref(stride)), "pt_stable_planes_plane_stride"); stride = int(stride.value)
        out = dict(header=np.zeros((4, h, w), np.uint32), planes=np.zeros((3 * stride, 20), np.uint32), stable_radiance=np.zeros((h, w, 4), np.uint16), depth=np.zeros((h, w), np.float32),
                   spec_hit_t=np.zeros((h, w), np.float32), motion_vectors=np.zeros((h, w, 4), np.uint16), throughput=np.zeros((h, w), np.uint32))
        g = self.L.pt_get_stable_planes; g.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t] + [ctypes.c_void_p] * 5; g.restype = ctypes.c_int32
        self._chk(g(self.h, _p(out["header"]), _p(out["planes"]), 3 * stride, _p(out["stable_radiance"]), _p(out["depth"]), _p(out["spec_hit_t"]), _p(out["motion_vectors"]), _p(out["throughput"])), "pt_get_stable_planes")
        out["plane_stride"] = stride
        return out

    def stable_planes_shard_bytes(self, rank):
        f = self.L.pt_stable_planes_shard_bytes; f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]; f.restype = ctypes.c_int32
        n = ctypes.c_size_t(0); self._chk(f(self.h, int(rank), ctypes.byref(n)), "pt_stable_planes_shard_bytes"); return int(n.value)

    def pack_stable_planes(self, device_ptr, nbytes):
        f = self.L.pt_pack_stable_planes; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]; f.restype = ctypes.c_int32
        self._chk(f(self.h, ctypes.c_void_p(device_ptr), nbytes), "pt_pack_stable_planes")

    def unpack_stable_planes(self, device_ptr, nbytes, rank):
        f = self.L.pt_unpack_stable_planes; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]; f.restype = ctypes.c_int32
        self._chk(f(self.h, ctypes.c_void_p(device_ptr), nbytes, int(rank)), "pt_unpack_stable_planes")

    def pack_stable_plane_guides(self, device_ptr, nbytes):
        """pt_pack_stable_plane_guides: depth | specular hit distance | motion vectors of this rank's pixels, 16 bytes each (what the other ranks' bakers need of the build pass)"""
        f = self.L.pt_pack_stable_plane_guides; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]; f.restype = ctypes.c_int32
        self._chk(f(self.h, ctypes.c_void_p(device_ptr), nbytes), "pt_pack_stable_plane_guides")

    def unpack_stable_plane_guides(self, device_ptr, nbytes, rank):
        f = self.L.pt_unpack_stable_plane_guides; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]; f.restype = ctypes.c_int32
        self._chk(f(self.h, ctypes.c_void_p(device_ptr), nbytes, int(rank)), "pt_unpack_stable_plane_guides")

    def neeat_update_begin(self):
        """pt_neeat_update_begin: LightsBaker::UpdateBegin on its own (the realtime frame in parts)"""
        f = self.L.pt_neeat_update_begin; f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int32; self._chk(f(self.h), "pt_neeat_update_begin")

    def neeat_update_end(self):
        """pt_neeat_update_end: LightsBaker::UpdateEnd on the depth and motion vectors the last build_stable_planes left"""
        f = self.L.pt_neeat_update_end; f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int32; self._chk(f(self.h), "pt_neeat_update_end")

    def gather_stable_planes(self):
        f = self.L.pt_gather_stable_planes; f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int32
        self._chk(f(self.h), "pt_gather_stable_planes")

    def realtime_frame(self, sample_index, params):
        """pt_realtime_frame: UpdateBegin -> build pass -> UpdateEnd on this frame's depth and motion vectors -> params.subSampleCount fill passes (with pt_set_neeat: the baker in the loop).
        Returns (buffers as get_stable_planes, build stats, fill stats)."""
        prm = np.ascontiguousarray(params); assert prm.dtype.itemsize == 224
        f = self.L.pt_realtime_frame; f.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]; f.restype = ctypes.c_int32
        b, fl = PtFrameStats(), PtFrameStats()
        self._chk(f(self.h, int(sample_index), _p(prm), ctypes.byref(b), ctypes.byref(fl)), "pt_realtime_frame")
        return self.get_stable_planes(), b.as_dict(), fl.as_dict()

    def stable_planes_merge(self):
        """pt_stable_planes_merge: the realtime frame without a denoiser (stable + noisy radiance) into the radiance buffer; returns radiance()"""
        f = self.L.pt_stable_planes_merge; f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int32
        self._chk(f(self.h), "pt_stable_planes_merge")
        return self.radiance()

    def denoise_spec_hit_t(self):
        """pt_denoise_spec_hit_t: the fill-in of the specular hit distance that ends a realtime frame's noisy passes; returns spec_hit_t [h, w] f32"""
        f = self.L.pt_denoise_spec_hit_t; f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int32
        self._chk(f(self.h), "pt_denoise_spec_hit_t")
        out = np.zeros((self.height, self.width), np.float32)
        g = self.L.pt_get_stable_planes; g.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t] + [ctypes.c_void_p] * 5; g.restype = ctypes.c_int32
        self._chk(g(self.h, None, None, 0, None, None, _p(out), None, None), "pt_get_stable_planes")
        return out

    def set_neeat(self, enable=True, global_feedback_weight=0.75, ratio=0.65, ssc_threshold=0.3, prefilter=True):
        """NEE-AT with the light baker in the loop (pt_set_neeat): every sample of render() becomes a frame — feedback passes, then the path tracer"""
        f = self.L.pt_set_neeat; f.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int32]; f.restype = ctypes.c_int32
        self._chk(f(self.h, 1 if enable else 0, float(global_feedback_weight), float(ratio), float(ssc_threshold), 1 if prefilter else 0), "pt_set_neeat")

    def neeat_reset(self): self._chk(self.L.pt_neeat_reset(self.h), "pt_neeat_reset")

    def neeat_tables(self):
        """(tile table uint32 [tilesY, tilesX, 128], jitter (x, y)) the last frame was traced with"""
        txy = np.zeros(2, np.uint32); jxy = np.zeros(2, np.uint32)
        self._chk(self.L.pt_get_neeat_tables(self.h, _p(txy), _p(jxy), None, 0), "pt_get_neeat_tables")
        t = np.zeros((int(txy[1]), int(txy[0]), 128), np.uint32)
        self._chk(self.L.pt_get_neeat_tables(self.h, None, None, _p(t), t.size), "pt_get_neeat_tables")
        return t, (int(jxy[0]), int(jxy[1]))

    def neeat_pack_feedback(self, device_ptr, nbytes):
        self.L.pt_neeat_pack_feedback.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self._chk(self.L.pt_neeat_pack_feedback(self.h, ctypes.c_void_p(device_ptr), nbytes), "pt_neeat_pack_feedback")

    def neeat_unpack_feedback(self, device_ptr, nbytes, rank):
        self.L.pt_neeat_unpack_feedback.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        self._chk(self.L.pt_neeat_unpack_feedback(self.h, ctypes.c_void_p(device_ptr), nbytes, rank), "pt_neeat_unpack_feedback")

    def light_feedback(self, sample=0):
        """the feedback reservoirs sample `sample` of the last render() call filled: (total weight float32 [H, W], candidate uint32 [H, W])"""
        w = np.zeros((self.height, self.width), np.float32); c = np.zeros((self.height, self.width), np.uint32)
        self._chk(self.L.pt_get_light_feedback(self.h, int(sample), _p(w), _p(c)), "pt_get_light_feedback")
        return w, c

    def animate(self, instances=None, positions=None, rebuild=False, vertex_ranges=None):
        """pt_animate; vertex_ranges = [(first, count), ...]: pt_animate_ranges — only those vertices of `positions` moved"""
        if vertex_ranges is None:
            self._chk(self.L.pt_animate(self.h, _p(instances), 0 if instances is None else len(instances), _p(positions), 0 if positions is None else positions.shape[0],
                                        1 if rebuild else 0), "pt_animate")
            return
        f = self.L.pt_animate_ranges; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32]; f.restype = ctypes.c_int32
        r = np.ascontiguousarray(np.asarray(vertex_ranges, np.uint32).reshape(-1, 2))
        self._chk(f(self.h, _p(instances), 0 if instances is None else len(instances), _p(positions), 0 if positions is None else positions.shape[0], _p(r), len(r), 1 if rebuild else 0), "pt_animate_ranges")

    def set_motion_history(self, enable=True):
        """pt_set_motion_history: every animate() call is a scene refresh from now on (the pose it finds becomes the previous pose); the stable-plane build pass's motion vectors carry object motion"""
        f = self.L.pt_set_motion_history; f.argtypes = [ctypes.c_void_p, ctypes.c_int32]; f.restype = ctypes.c_int32
        self._chk(f(self.h, 1 if enable else 0), "pt_set_motion_history")

    def set_previous_pose(self, instances=None, positions=None):
        """pt_set_previous_pose: the previous frame's instance transforms / vertex positions handed over directly (None = that part did not move)"""
        f = self.L.pt_set_previous_pose; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]; f.restype = ctypes.c_int32
        a = None if instances is None else np.ascontiguousarray(instances); b = None if positions is None else np.ascontiguousarray(positions, np.float32)
        self._chk(f(self.h, _p(a), 0 if a is None else len(a), _p(b), 0 if b is None else b.shape[0]), "pt_set_previous_pose")

    def animate_normals(self, normals=None, tangents=None):
        """pt_animate_normals: the deformed meshes' packed vertex normals / tangents (uint32 per vertex, SNORM8) — with pt_animate(positions) a skinned frame is complete"""
        f = self.L.pt_animate_normals; f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]; f.restype = ctypes.c_int32
        n = len(normals) if normals is not None else len(tangents)
        a = None if normals is None else np.ascontiguousarray(normals, np.uint32); b = None if tangents is None else np.ascontiguousarray(tangents, np.uint32)
        self._chk(f(self.h, _p(a), _p(b), n), "pt_animate_normals")

    # ---- per-frame state
    def set_camera(self, cam):
        cam = np.ascontiguousarray(cam)
        self._chk(self.L.pt_set_camera(self.h, _p(cam)), "pt_set_camera")

    def default_settings(self):
        s = np.zeros((), dtype=scenes.SETTINGS_DTYPE)
        self._chk(self.L.pt_default_settings(_p(s)), "pt_default_settings")
        return s

    def set_settings(self, s):
        s = np.ascontiguousarray(s)
        self._chk(self.L.pt_set_settings(self.h, _p(s)), "pt_set_settings")

    def resize(self, w, h):
        self.width, self.height = w, h
        self._chk(self.L.pt_resize(self.h, w, h), "pt_resize")

    def reset_accumulation(self):
        self._chk(self.L.pt_reset_accumulation(self.h), "pt_reset_accumulation")

    def set_counters(self, enable):
        self._chk(self.L.pt_set_counters(self.h, 1 if enable else 0), "pt_set_counters")

    def render(self, first, count):
        st = PtFrameStats()
        self._chk(self.L.pt_render(self.h, first, count, ctypes.byref(st)), "pt_render")
        return st.as_dict()

    def radiance(self):
        ptr = ctypes.POINTER(ctypes.c_float)()
        pitch = ctypes.c_size_t()
        self._chk(self.L.pt_map_radiance(self.h, ctypes.byref(ptr), ctypes.byref(pitch)), "pt_map_radiance")
        img = np.ctypeslib.as_array(ptr, shape=(self.height, self.width, 4)).copy()
        self.L.pt_unmap_radiance(self.h)
        return img

    def set_serial_kernels(self, enable):
        self._chk(self.L.pt_set_serial_kernels(self.h, 1 if enable else 0), "pt_set_serial_kernels")

    def set_tail_paths(self, max_paths):
        """pt_set_tail_paths: batches with at most this many live paths are finished by the tail kernel (0: never)."""
        self._chk(self.L.pt_set_tail_paths(self.h, int(max_paths)), "pt_set_tail_paths")

    def set_fused_traversal(self, mode):
        """pt_set_fused_traversal: 0 = visibility rays in launches of their own, 1 = in the next bounce's closest-hit launch (k_trace_pair), 2 = by the size of the call (default)."""
        self._chk(self.L.pt_set_fused_traversal(self.h, int(mode)), "pt_set_fused_traversal")

    def tonemap(self, params=None):
        """pt_tonemap: the accumulation buffer through ToneMappingPass into sRGB RGBA8 -> (H, W, 4) uint8."""
        t = default_tonemap() if params is None else params
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self.L.pt_tonemap.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self._chk(self.L.pt_tonemap(self.h, _p(t), _p(out), out.nbytes), "pt_tonemap")
        return out

    # ---- multi-GPU shard plumbing (device pointers come from torch tensors)
    def average_luminance(self):
        """pt_average_luminance: the auto-exposure luminance capture (ToneMappingPasses.cpp:225-288) of the current accumulation buffer."""
        v = ctypes.c_float(0.0)
        self.L.pt_average_luminance.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        self._chk(self.L.pt_average_luminance(self.h, ctypes.byref(v)), "pt_average_luminance")
        return float(v.value)

    def shard_info(self):
        n, b = ctypes.c_uint32(), ctypes.c_size_t()
        self._chk(self.L.pt_shard_info(self.h, ctypes.byref(n), ctypes.byref(b)), "pt_shard_info")
        return n.value, b.value

    def pack_shard(self, device_ptr, nbytes):
        self._chk(self.L.pt_pack_shard(self.h, ctypes.c_void_p(device_ptr), ctypes.c_size_t(nbytes)), "pt_pack_shard")

    def unpack_shard(self, device_ptr, nbytes, rank):
        self._chk(self.L.pt_unpack_shard(self.h, ctypes.c_void_p(device_ptr), ctypes.c_size_t(nbytes), rank), "pt_unpack_shard")

    # ---- probes
    def comm_init(self, unique_id, rank, world):
        """pt_comm_init (collective): the RCCL communicator of the frame gather on this context's device."""
        _share_rccl_with_torch()
        buf = (ctypes.c_ubyte * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._chk(self.L.pt_comm_init(self.h, buf, rank, world), "pt_comm_init")

    def gather(self):
        """pt_gather (collective): every rank's tiles -> rank 0's accumulation buffer, on the library's stream."""
        self._chk(self.L.pt_gather(self.h), "pt_gather")

    def comm_destroy(self):
        self._chk(self.L.pt_comm_destroy(self.h), "pt_comm_destroy")

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.zeros((rays.shape[0], 4), np.float32)
        ms = ctypes.c_double()
        self._chk(self.L.pt_trace_closest(self.h, _p(rays), rays.shape[0], _p(out), ctypes.byref(ms)), "pt_trace_closest")
        return out, ms.value

    def trace_visibility(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.zeros(rays.shape[0], np.uint32)
        ms = ctypes.c_double()
        self._chk(self.L.pt_trace_visibility(self.h, _p(rays), rays.shape[0], _p(out), ctypes.byref(ms)), "pt_trace_visibility")
        return out, ms.value

    def lights(self):
        n, npx, dim = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._chk(self.L.pt_get_lights(self.h, ctypes.byref(n), ctypes.byref(npx), None, None, None, None, None, ctypes.byref(dim)), "pt_get_lights")
        lights = np.zeros((n.value, 8), np.uint32)
        ex = np.zeros((n.value, 4), np.uint32)
        pc = np.zeros(n.value, np.uint32)
        pi = np.zeros(npx.value, np.uint32)
        el = np.zeros(dim.value * dim.value, np.uint32)
        self._chk(self.L.pt_get_lights(self.h, None, None, _p(lights), _p(ex), _p(pc), _p(pi), _p(el), None), "pt_get_lights")
        return dict(lights=lights, lightsEx=ex, proxyCounters=pc, proxyIndices=pi, envLookup=el, envLookupDim=dim.value)

    def env_cube(self):
        """pt_get_env_cube: the baked RGBA16F environment cube as uint32 [texels, 2] plus (dim, mipLevels)."""
        n, dim, lv = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._chk(self.L.pt_get_env_cube(self.h, ctypes.byref(n), ctypes.byref(dim), ctypes.byref(lv), None, 0), "pt_get_env_cube")
        out = np.zeros((n.value, 2), np.uint32)
        if n.value: self._chk(self.L.pt_get_env_cube(self.h, None, None, None, _p(out), n.value), "pt_get_env_cube")
        return out, dim.value, lv.value

    def subinstances(self):
        n = ctypes.c_uint32()
        self._chk(self.L.pt_get_subinstances(self.h, ctypes.byref(n), None), "pt_get_subinstances")
        out = np.zeros((n.value, 8), np.uint32)
        self._chk(self.L.pt_get_subinstances(self.h, None, _p(out)), "pt_get_subinstances")
        return out

    def scene_info(self):
        a, b, c, d = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._chk(self.L.pt_get_scene_info(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)), "pt_get_scene_info")
        return dict(triangles=a.value, bvhNodes=b.value, instances=c.value, materials=d.value)

    def build_stats(self):
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._chk(self.L.pt_get_build_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "pt_get_build_stats")
        return dict(buildMs=a.value, refitMs=b.value, lightBakeMs=c.value)

    def bvh_info(self):
        """pt_get_bvh_info: which builder made the tree and where it ran."""
        class PtBvhInfo(ctypes.Structure):
            _fields_ = [("builder", ctypes.c_uint32), ("builtOnDevice", ctypes.c_uint32), ("numTriangles", ctypes.c_uint32), ("numWideNodes", ctypes.c_uint32),
                        ("collapseLevels", ctypes.c_uint32), ("optimiserPasses", ctypes.c_uint32), ("hostMs", ctypes.c_float), ("buildMs", ctypes.c_float)]
        o = PtBvhInfo()
        self._chk(self.L.pt_get_bvh_info(self.h, ctypes.byref(o)), "pt_get_bvh_info")
        names = {0: "PLOC", 1: "Karras radix tree", 2: "binned SAH + re-insertion (host)", 3: "PLOC + parallel re-insertion + cost-driven wide nodes"}
        d = {k: getattr(o, k) for k, _ in PtBvhInfo._fields_}
        d["builderName"] = names.get(o.builder, "?"); d["builtOn"] = "device" if o.builtOnDevice else "host"
        return d

    def probe(self, kind, inp, out_shape, out_dtype=np.float32):
        """pt_probe (include/mi355pt_testhooks.h): only on a PathTracer(test_hooks=True)"""
        if not self.test_hooks: raise RuntimeError("pt_probe is not part of the shipped library: PathTracer(test_hooks=True)")
        inp = np.ascontiguousarray(inp)
        out = np.zeros(out_shape, out_dtype)
        self._chk(self.L.pt_probe(self.h, kind, _p(inp), inp.nbytes, _p(out), out.nbytes, out_shape[0]), "pt_probe")
        return out
