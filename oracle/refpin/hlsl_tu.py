#!/usr/bin/env python3
"""ORACLE pin (test infrastructure only): writes to stdout ONE C++ translation unit made of
     oracle/refpin/hlsl_shim.h  +  functions taken verbatim from the reference's .hlsli files (read where they lie under
     /root/reference, streamed straight into the compiler: no reference text is ever written into this repo)  +  hlsl_wrappers.inc
   oracle/Makefile pipes it into g++ (-fsingle-precision-constant: HLSL literals are fp32) -> oracle/_ref/librefpin_hlsl.so.

The only edits made to the reference text are the ones C++ syntax forces:
  * `out T x` / `inout T x` parameters become `T& x`; the `in` qualifier is dropped
  * vector swizzles `.xy` `.yx` `.xyz` `.rgb` ... become member calls `.xy_()` (hlsl_shim.h); swizzles on scalars (`_alpha.xx`, `(expr).xxx`,
    `packedData.x`) become constructor calls / the scalar itself
  * `const` is dropped from by-value parameters (HLSL methods are not const-qualified); `this.` becomes `this->`; `: register(...)`, `row_major`,
    `precise` and leading `::` are dropped; untyped `Texture2D` becomes `Texture2D<float4>`
  * `[unroll]`-style attributes and the `uniform` parameter qualifier are dropped
usage: hlsl_tu.py /root/reference > tu.cpp"""
import os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHADERS = "Rtxpt/Shaders/PathTracer"

# (file, what): what = list of items taken from that file, in emission order:
#   "name"            every file-scope function definition of that name (overloads included)
#   "#K_"             every `#define K_*` of the file, as a float constant
#   "pp"              every preprocessor line of the file except #include (configuration headers)
#   "struct Name"     that struct / enum class definition; "struct Name -method" drops a member function (one that needs types outside the pin)
#   "range A..B"      the file from the first line matching regex A up to (not including) the first later line matching regex B
#   "text ..."        emitted as is (namespace brackets, macro switches)
PLAN = [
    ("Utils/Math/MathConstants.hlsli", ["#K_"]),
    ("Utils/Utils.hlsli", ["FastSqrt", "FastACos", "Luminance", "Average"]),
    ("Rendering/Materials/Fresnel.hlsli", ["evalFresnelSchlick", "evalFresnelDielectric"]),
    ("Rendering/Materials/Microfacet.hlsli", ["evalNdfGGX", "evalPdfGGX_BVNDF", "sampleGGX_BVNDF", "evalLambdaGGX", "evalMaskingSmithGGXCorrelated"]),
    ("Utils/Math/MathHelpers.hlsli", ["ndir_to_oct_equal_area_unorm", "oct_to_ndir_equal_area_unorm", "sample_disk", "sample_disk_concentric",
                                      "sample_cosine_hemisphere_concentric", "perp_stark"]),
    ("PathTracerHelpers.hlsli", ["ComputeRayOrigin", "ComputeLowGrazingAngleFalloff", "ComputeRayConeSpreadAngleExpansionByScatterPDF",
                                 "ComputeNewScatterFireflyFilterK", "FireflyFilter", "FireflyFilterShort"]),
    # ---- the stateless sample generators (integer RNG streams: must match bit for bit)
    ("Utils/NoiseAndSequences.hlsli", ["range #define\\s+SOBOL_MAX_DIMENSIONS..^\\s*$", "Hash32", "Hash32Combine", "Hash32ToFloat", "bhos_sobol", "bhos_reverse_bits", "bhos_owen_hash", "bhos_owen_scramble"]),
    ("Utils/SampleGenerators.hlsli", ["struct SampleGeneratorEffectSeed"]),
    ("Utils/StatelessSampleGenerators.hlsli", ["struct SampleGeneratorVertexBase", "struct SampleSequenceGenerator", "struct UniformSampleSequenceGenerator"]),
    ("Utils/SampleGenerators.hlsli", ["sampleNext1D"]),
    # ---- polymorphic lights: packing, sphere / triangle / environment-quad sampling and MIS pdfs
    ("Utils/Utils.hlsli", ["sq", "OctWrap", "Encode_Oct", "Decode_Oct", "NDirToOctUnorm32", "OctToNDirUnorm32"]),
    ("Utils/Geometry.hlsli", ["BranchlessONB", "SampleTriangleUniform", "pdfAtoW"]),
    ("Utils/Packing.hlsli", ["range #define PACK_UFLOAT_TEMPLATE..^uint Pack_R11G11B10"]),
    ("Lighting/PolymorphicLight.h", ["range #define DISTANT_LIGHT_DISTANCE..#endif // __POLYMORPHIC_LIGHT_H__"]),
    ("Lighting/PolymorphicLightPTConfig.h", ["pp"]),
    ("Lighting/LightShaping.hlsli", ["range ^struct LightShaping..#endif // LIGHT_SHAPING_HLSLI"]),
    ("Lighting/PolymorphicLight.hlsli", ["range #define FLT_EPSILON_MINI..#endif // __POLYMORPHIC_LIGHT_HLSLI__ -Eval"]),
    # ---- light baking, the per-light / per-node functions of the LightsBaker compute passes (the passes themselves use group-shared memory and atomics)
    ("Lighting/LightingConfig.h", ["text #define STATIC_ASSERT(X)", "range ^#define RTXPT_LIGHTING_MAX_LIGHTS..^#endif // #define __LIGHTING_CONFIG_H__"]),
    ("@Rtxpt/Lighting/LightsBaker.hlsl", ["text struct PinEnvMapParams { float3 ColorMultiplier; }; struct PinBakerConsts { uint EnvMapImportanceMapMIPCount; PinEnvMapParams EnvMapParams; float DistantVsLocalRelativeImportance; };",
                                          "text struct PinImportanceMap { const float4* const* mips; const uint* dims; float4 Load(int3 c) const { return mips[c.z][(uint)c.y * dims[c.z] + (uint)c.x]; } };",
                                          "text static PinBakerConsts g_bakerConsts; static PinImportanceMap t_envRadianceAndImportanceMap;",
                                          "EnvironmentComputeRadianceAndWeight", "range ^#define PACK_20F_12UI..^uint EnvironmentComputeWeightForQTBuild", "EnvironmentComputeWeightForQTBuild",
                                          "EQTNodePack", "EQTNodeUnpack", "ComputeWeight"]),
    # ---- display path: ToneMapping.ps.hlsli whole, over a colour "texture" that holds one pixel
    ("@Rtxpt/ToneMapper/ToneMapping_cb.h", ["range ^#define TONEMAPPING_AUTOEXPOSURE_CPU..^#endif // TONEMAPPING_CB_H"]),
    ("@Rtxpt/ToneMapper/ToneMapping.ps.hlsli", ["text struct PinColorTexture { float4 texel; float4 Sample(SamplerState, float2) const { return texel; } float4 SampleLevel(SamplerState, float2, float) const { return texel; } };",
                                                "text static PinColorTexture gColorTex, gLuminanceTex; static SamplerState gColorSampler, gLuminanceTexSampler;",
                                                "range ^static const float kExposureKey..^#endif //__TONE_MAPPING_PS_HLSLI__"]),
    # ---- the whole standard BSDF (FalcorBSDF and its four lobes), once per diffuse model
    ("Utils/Math/MathConstants.hlsli", ["range static const float\\s+cFloatOneMinusEpsilon..^\\s*$"]),
    ("Rendering/Materials/LobeType.hlsli", ["struct LobeType"]),
    ("Scene/Material/MaterialData.hlsli", ["range #define EXTRACT_BITS..^struct", "struct MaterialHeader"]),
    ("Scene/ShadingData.hlsli", ["struct ShadingData"]),
    ("Rendering/Materials/BxDFConfig.hlsli", ["text #define DiffuseBrdf 0", "pp", "text namespace lambert {"]),
    ("Rendering/Materials/BxDF.hlsli", ["range static const float kMinCosTheta..^struct FalcorBSDF", "struct FalcorBSDF -evalDeltaLobes", "text } // lambert"]),
    ("Rendering/Materials/BxDFConfig.hlsli", ["text #undef DiffuseBrdf", "text #define DiffuseBrdf 2", "text namespace frostbite {"]),
    ("Rendering/Materials/BxDF.hlsli", ["range static const float kMinCosTheta..^struct FalcorBSDF", "struct FalcorBSDF -evalDeltaLobes", "text } // frostbite"]),
]
SKIP_SIGNATURE = re.compile(r"\bhalf\d?\b|\bmin16\w+|\bfloat16_t\d?\b")       # overloads in types the fp32 build never uses


def strip_comments(text):
    out, i, n = [], 0, len(text)
    while i < n:
        if text.startswith("//", i):
            j = text.find("\n", i); j = n if j < 0 else j
            i = j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2); j = n - 2 if j < 0 else j
            out.append("\n" * text.count("\n", i, j + 2)); i = j + 2
        else:
            out.append(text[i]); i += 1
    return "".join(out)


def extract_function(text, name, path):
    """every definition `<type> name(<params>) {...}` at file scope, overloads included"""
    found, seen = [], set()
    for m in re.finditer(r"^[ \t]*((?:static\s+|inline\s+|const\s+)*[A-Za-z_]\w*)\s+" + re.escape(name) + r"\s*\(([^)]*)\)\s*(?:const\s*)?\{", text, re.M):
        depth, i = 0, m.end() - 1
        while True:
            c = text[i]
            if c == "{": depth += 1
            elif c == "}":
                depth -= 1
                if depth == 0: break
            i += 1
        body = text[m.start():i + 1]
        head = text[:m.start()].rstrip().rsplit("\n", 1)[-1]
        if re.match(r"\s*template\s*<", head): body = head + "\n" + body      # function templates: the template line sits above the signature
        if SKIP_SIGNATURE.search(m.group(0)): continue
        sig = re.sub(r"\s+", " ", m.group(0).replace("lpfloat", "float"))
        if sig in seen: continue                      # the lpfloat overload of an fp32 build is the same function twice
        seen.add(sig); found.append(body)
    if not found: raise SystemExit("hlsl_tu.py: %s not found in %s" % (name, path))
    return found


def match_brace(text, i):
    depth = 0
    while True:
        c = text[i]
        if c == "{": depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0: return i
        i += 1


def extract_struct(text, spec, path):
    parts = spec.split()
    name, drop = parts[0], [p[1:] for p in parts[1:] if p.startswith("-")]
    m = re.search(r"^[ \t]*(?:struct|class|enum class|enum)\s+" + re.escape(name) + r"\b[^{;]*\{", text, re.M)
    if not m: raise SystemExit("hlsl_tu.py: struct %s not found in %s" % (name, path))
    end = match_brace(text, m.end() - 1)
    body = text[m.start():end + 1] + ";"
    for d in drop:
        k = re.search(r"^[ \t]*[A-Za-z_][\w<>]*\s+" + re.escape(d) + r"\s*\([^)]*\)\s*\{", body, re.M)
        if not k: raise SystemExit("hlsl_tu.py: member %s of %s not found in %s" % (d, name, path))
        body = body[:k.start()] + body[match_brace(body, k.end() - 1) + 1:]
    return body


def extract_range(text, spec, path, raw=None):
    drop = [w[1:] for w in spec.split() if w.startswith("-")]
    spec = " ".join(w for w in spec.split(" ") if not (w.startswith("-") and w[1:] in drop))
    a, b = spec.split("..", 1)
    lines = text.split("\n")
    marks = (raw if raw is not None else text).split("\n")             # markers may sit in comments; comment stripping keeps the line structure
    assert len(marks) == len(lines)
    i0 = next((i for i, l in enumerate(marks) if re.search(a, l)), None)
    if i0 is None: raise SystemExit("hlsl_tu.py: range start /%s/ not found in %s" % (a, path))
    i1 = next((i for i in range(i0 + 1, len(marks)) if re.search(b, marks[i])), None)
    if i1 is None: raise SystemExit("hlsl_tu.py: range end /%s/ not found in %s" % (b, path))
    body = "\n".join(l for l in lines[i0:i1] if not re.match(r"\s*#\s*include", l))
    for d in drop:                                     # functions that need declarations outside the pin
        while True:
            k = re.search(r"^[ \t]*[A-Za-z_][\w<>]*\s+" + re.escape(d) + r"\s*\([^)]*\)\s*\{", body, re.M)
            if not k: break
            body = body[:k.start()] + body[match_brace(body, k.end() - 1) + 1:]
    return body


def to_cpp(code):
    # parameters
    code = re.sub(r"\b(?:inout|out)\s+(const\s+)?([A-Za-z_][\w:]*)\s+([A-Za-z_]\w*)\s*\[([^\]]+)\]", lambda m: "%s%s (&%s)[%s]" % (m.group(1) or "", m.group(2), m.group(3), m.group(4)), code)
    code = re.sub(r"\b(?:inout|out)\s+(const\s+)?([A-Za-z_][\w:]*(?:\s*<[^<>]*>)?)\s+([A-Za-z_]\w*)", lambda m: "%s%s& %s" % (m.group(1) or "", m.group(2), m.group(3)), code)
    code = re.sub(r"([(,]\s*(?:const\s+)?)in\s+(?=(?:const\s+)?[A-Za-z_]\w*\s+[A-Za-z_]\w*)", r"\1", code)
    code = re.sub(r"\bconst\s+(?=[A-Za-z_][\w:]*\s+[A-Za-z_]\w*\s*[,)])", "", code)        # by-value parameters: HLSL calls non-const methods on them
    code = re.sub(r"\bconst\s+(?=[A-Z]\w*\s+[A-Za-z_]\w*\s*=)", "", code)                  # `const Struct local = ...` likewise
    code = re.sub(r"\bthis\.", "this->", code)
    # `float2(NextFloat(), NextFloat())`: HLSL evaluates constructor arguments left to right, C++ leaves the order open (gcc: right to left)
    code = code.replace("return float2(NextFloat(), NextFloat());", "{ float nf0 = NextFloat(); float nf1 = NextFloat(); return float2(nf0, nf1); }")
    # `cond ? float : lpfloat`: HLSL promotes, C++ wants one type (no-ops in the fp32 build)
    code = code.replace("alpha < kMinGGXAlpha ? 0.f : dataRoughness", "alpha < kMinGGXAlpha ? 0.f : (float)dataRoughness")
    code = code.replace("(applyMIS)?(path.GetBsdfScatterPdf()):(0.0)", "(applyMIS)?((float)path.GetBsdfScatterPdf()):(0.0)")
    # swizzles on scalars (literals, named scalars, parenthesised / call expressions) become constructor calls ...
    code = re.sub(r"(?<![\w.])(\d+\.\d*f?|\.\d+f?|\d+)\.(xx|xxx|xxxx)\b", lambda m: "float%d(%s)" % (len(m.group(2)), m.group(1)), code)     # `0.5.xx`, `0.xxx`
    code = code.replace("1.#INF", "__builtin_inff()")                                           # MSVC-style infinity literal (PathTracerStablePlanes.hlsli)
    code = code.replace("DeltaLobe deltaLobes[cMaxDeltaLobes]; uint deltaLobeCount; float nonDeltaPart;", "DeltaLobe deltaLobes[cMaxDeltaLobes]; int deltaLobeCount; float nonDeltaPart;")      # passed to an `out int` parameter
    code = re.sub(r"\b(HLF_MAX|kNRDMinReflectance|kNRDMaxReflectance|_alpha|radiance|unpackedRadiance|offset|index|width|RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE|fp16Max|destinationRes|cubeDim|invSamples)\.(xx|xxx|xxxx)\b", lambda m: "float%d(%s)" % (len(m.group(2)), m.group(1)), code)
    code = re.sub(r"\b((?:\w+\.)?AttenuationDistance)\.(xx|xxx|xxxx)\b", lambda m: "float%d(%s)" % (len(m.group(2)), m.group(1)), code)
    code = re.sub(r"((?:\b[A-Za-z_][\w.]*)?\((?:[^()]|\((?:[^()]|\([^()]*\))*\))*\))\.(xx|xxx|xxxx|rr|rrr|rrrr)\b", lambda m: "float%d(%s)" % (len(m.group(2)), m.group(1)), code)
    code = re.sub(r"\bpackedData\.x\b", "packedData", code)                               # `.x` of a scalar
    # ... swizzles on vectors become member calls
    code = re.sub(r"\.(xy|yx|xx|xz|yz|zw|xyz|rgb|rgba|xyw|xzw|yzw|xyzw|wzyx)\b", r".\1_()", code)
    code = re.sub(r"\(\s*([A-Z]\w*)\s*\)\s*0\b(?!\.)", r"\1()", code)                        # `(Struct)0`: zero initialisation
    code = re.sub(r"\[(?:unroll|loop|branch|flatten|mutating|forceinline)(?:\([^)]*\))?\][ \t]*", "", code)
    code = re.sub(r"\buniform\s+(?=uint|int|float|bool)", "", code)
    code = re.sub(r":\s*register\s*\([^)]*\)", "", code)                                     # resource bindings
    code = re.sub(r"\s*:\s*SV_\w+", "", code)                                                # system-value semantics of entry-point parameters
    code = re.sub(r"\b(Texture2D|TextureCube|RWTexture2D|RWTexture2DArray)\b(?!\s*<)", r"\1<float4>", code)         # untyped resource = float4 elements
    code = re.sub(r"\b(row_major|precise|nointerpolation|globallycoherent)[ \t]+", "", code)
    code = re.sub(r"\bcbuffer\s+\w+\s*\{([^{}]*)\}\s*;?", r"static \1", code)                  # a constant buffer is its members, as globals
    code = re.sub(r"(?<![\w:>)])::(?=[A-Za-z_])", "", code)                                  # `::name` (global scope): everything lives in one namespace here
    return code


# ---- second translation unit: the integrator. Whole files, in the order the preprocessor would visit them (includes are resolved here so that
# every file's text goes through to_cpp); files on the deny list are replaced by the stubs of hlsl_pt_stubs.h.
PT_ROOTS = ["../PathTracerBridge.hlsli", "PathTracer.hlsli"]
PT_DENY = ("ShaderDebug.hlsl", "PathTracerDebug.hlsli", "STFSamplerState.hlsli", "Xoshiro.hlsli", "SplitMix64.hlsli", "BitTricks.hlsli",
           "HitInfo.hlsli", "HitInfoType.hlsli", "PackedFormats.hlsli", "FormatConversion.hlsli", "ColorHelpers.hlsli", "Quaternion.hlsli", "SceneTypes.hlsli")
# files of which only some items are needed (the rest uses matrix member syntax / half types / pixel-shader intrinsics that the pin has no use for)
PT_PICK = {
    "MathHelpers.hlsli": ["ndir_to_oct_equal_area_unorm", "oct_to_ndir_equal_area_unorm", "sample_disk", "sample_disk_concentric", "sample_cosine_hemisphere_concentric",
                          "sample_cosine_hemisphere_polar", "perp_stark", "sqr", "world_to_latlong_map"],
}


def emit_file(path, w, done):
    path = os.path.normpath(path)
    if path in done: return
    done.add(path)
    raw = open(path, encoding="latin-1").read()
    text = strip_comments(raw)
    if os.path.basename(path) in PT_PICK:
        w("// ======== %s (selected items)\n" % os.path.basename(path))
        for name in PT_PICK[os.path.basename(path)]:
            for body in extract_function(text, name, path): w(to_cpp(body) + "\n")
        return
    text = re.sub(r"!\s*defined\s*\(\s*__cplusplus\s*\)", "1", text)
    text = re.sub(r"defined\s*\(\s*__cplusplus\s*\)", "0", text)
    text = re.sub(r"^([ \t]*)#\s*ifdef\s+__cplusplus\b", r"\1#if 0", text, flags=re.M)
    text = re.sub(r"^([ \t]*)#\s*ifndef\s+__cplusplus\b", r"\1#if 1", text, flags=re.M)
    w("// ======== %s\n" % os.path.relpath(path, os.path.join(REF, SHADERS)))
    chunk = []
    def flush():
        if chunk: w(to_cpp("\n".join(chunk)) + "\n"); del chunk[:]
    for line in text.split("\n"):
        m = re.match(r'\s*#\s*include\s*"([^"]+)"', line)
        if not m: chunk.append(line); continue
        flush()
        inc = os.path.join(os.path.dirname(path), m.group(1))
        if os.path.basename(inc) in PT_DENY or not os.path.exists(inc): w("// (not part of the pin: %s)\n" % m.group(1)); continue
        emit_file(inc, w, done)
    flush()


def main_pt(ref):
    global REF
    REF = ref
    w = sys.stdout.write
    w('// generated by oracle/refpin/hlsl_tu.py --integrator -- never written to disk\n#include "%s/hlsl_shim.h"\n#include "%s/../ptref/ptref_api.cpp"      // the oracle (scene services for the Bridge): before any reference macro exists\n#include "%s/hlsl_pt_stubs.h"\nnamespace hl {\n' % (HERE, HERE, HERE))
    done = set()
    for r in PT_ROOTS: emit_file(os.path.join(ref, SHADERS, r), w, done)
    # the RTXPT side of the Donut bridge (PathTracerBridgeDonut.hlsli): geometry fetch, material evaluation, loadSurface and the small accessors.
    # What it includes from Donut itself (un-vendored submodule) is restated in hlsl_pt_stubs.h ("donut side"); the ray-query functions stay with the driver.
    for extra in ("../PathTracer/Materials/MaterialPT.h", "../SubInstanceData.h", "../PathTracer/Materials/MaterialTypes.hlsli", "../Libraries/MicroRng.hlsli"):
        emit_file(os.path.join(ref, SHADERS, extra), w, done)
    w('#include "%s/hlsl_pt_bridge_stubs.h"\n' % HERE)
    bpath = os.path.join(ref, "Rtxpt/Shaders/PathTracerBridgeDonut.hlsli")
    braw = open(bpath, encoding="latin-1").read(); btext = strip_comments(braw)
    for spec in ("^enum DonutGeometryAttributes..^static OpacityMicroMapDebugInfo loadOmmDebugInfo",
                 "^uint Bridge::getSampleIndex..^// 2\\.5D motion vectors",
                 "^bool AlphaTestImpl..^bool Bridge::traceVisibilityRay",
                 "^EnvMap Bridge::CreateEnvMap..^void Bridge::ExportSurfaceInit",
                 "^float3 Bridge::computeMotionVector..^// 2\\.5D motion vectors",                 # (the stable-plane build pass calls it; zero in reference mode)
                 "^void Bridge::ExportSurfaceInit..^PathTracer::WorkingContext GetWorkingContext"):      # the guide-buffer dump: ExportSurfaceInit, ExportSurface, ExportNonSurface, ExportSpecHitTStart / Stop
        w("// ======== PathTracerBridgeDonut.hlsli : %s\n" % spec)
        w(to_cpp(extract_range(btext, spec, "PathTracerBridgeDonut.hlsli", braw)) + "\n")
    # the procedural sky (SampleProceduralSky.hlsli and, through its include, precomputed_sky.hlsli): whole files, ahead of the baker that calls them
    emit_file(os.path.join(ref, "Rtxpt/Lighting/Distant/SampleProceduralSky.hlsli"), w, done)
    # EnvMapBaker.hlsl: the cube bake (LowResPrePassLayerCS, BaseLayerCS, MIPReduceCS and everything they call) over the stand-in bindings of hlsl_envbake_stubs.h
    epath = os.path.join(ref, "Rtxpt/Lighting/Distant/EnvMapBaker.hlsl")
    etext = strip_comments(open(epath, encoding="latin-1").read())
    w("// ======== EnvMapBaker.hlsl (selected items)\nnamespace embake {\n" + re.search(r"^#define\s+EMB_MAXDIRLIGHTS\b.*$", etext, re.M).group(0) + "\n")
    w(to_cpp(extract_struct(etext, "EMB_DirectionalLight", "EnvMapBaker.hlsl")) + "\n")
    w('#include "%s/hlsl_envbake_stubs.h"\n' % HERE)
    for name in ("CubemapGetDirectionFor", "SampleSource", "SphereQuadrantArea", "CubemapTexelSolidAngle", "CubemapTexelSolidAngle4", "ComputeLightContribution", "GetProcSkyContext", "GenerateTexel",
                 "LowResPrePassLayerCS", "BaseLayerCS", "MIPReduceCS"):
        for body in extract_function(etext, name, "EnvMapBaker.hlsl"): w(to_cpp(body) + "\n")
    w("} // namespace embake\n")
    # EnvMapImportanceSamplingBaker.hlsl: the radiance / importance map pass the light baker's environment quad tree is built from
    ipath = os.path.join(ref, "Rtxpt/Lighting/Distant/EnvMapImportanceSamplingBaker.hlsl")
    itext = strip_comments(open(ipath, encoding="latin-1").read())
    w("// ======== EnvMapImportanceSamplingBaker.hlsl (selected items)\nnamespace emisb {\n")
    w(to_cpp(extract_struct(itext, "EnvMapImportanceSamplingBakerConstants", "EnvMapImportanceSamplingBaker.hlsl")) + "\n")
    w('#include "%s/hlsl_emisb_stubs.h"\n' % HERE)
    for body in extract_function(itext, "BuildMIPDescentImportanceMapCS", "EnvMapImportanceSamplingBaker.hlsl"): w(to_cpp(body) + "\n")
    w("} // namespace emisb\n")
    # BC6UCompress.hlsl: the cube compressor's one-region encoder (QUALITY 0 = "Fast", EnvMapBaker's default on D3D12), EncodeP1 and what it calls
    cpath = os.path.join(ref, "Rtxpt/Lighting/Distant/BC6UCompress.hlsl")
    ctext = strip_comments(open(cpath, encoding="latin-1").read())
    w("// ======== BC6UCompress.hlsl (selected items)\nnamespace bc6u {\n#define INSET_COLOR_BBOX 1\n#define OPTIMIZE_ENDPOINTS 1\n#define LUMINANCE_WEIGHTS 1\nstatic const float HALF_MAX = 65504.0f;\n")
    for name in ("CalcMSLE", "PatternFixupID", "Pattern", "Quantize7", "Quantize9", "Quantize10", "Unquantize7", "Unquantize9", "Unquantize10", "FinishUnquantize", "Swap", "ComputeIndex3", "ComputeIndex4", "SignExtend",
                 "InsetColorBBoxP1", "OptimizeEndpointsP1", "OptimizeEndpointsP2", "EncodeP1", "DistToLineSq", "EvaluateP2Pattern", "EncodeP2Pattern"):
        for body in extract_function(ctext, name, "BC6UCompress.hlsl"): w(to_cpp(body) + "\n")
    w("} // namespace bc6u\n")
    # LightsBaker.hlsl: the NEE-AT feedback passes (last frame's reservoirs -> this frame's tile tables and usage counts) over the stand-in bindings of hlsl_lbfb_stubs.h
    lpath = os.path.join(ref, "Rtxpt/Lighting/LightsBaker.hlsl")
    ltext = strip_comments(open(lpath, encoding="latin-1").read())
    ltext = re.sub(r"(?<![\w.])([01])\.xx\b", r"int2(\1,\1)", ltext)      # `0.xx` / `1.xx` next to int2 operands: integer splats (to_cpp would make them float2)
    ltext = re.sub(r"\b(RTXPT_NEEAT_EARLY_FEEDBACK_TILE_SIZE|RTXPT_LIGHTING_SAMPLING_BUFFER_TILE_SIZE)\.xx\b", r"int2(\1,\1)", ltext)
    w("// ======== LightsBaker.hlsl (NEE-AT feedback passes)\nnamespace lbfb {\n")
    w('#include "%s/hlsl_lbfb_stubs.h"\n' % HERE)
    for name in ("RemapPastToCurrent", "DistanceFromFrustum", "ImportanceBooster"):
        for body in extract_function(ltext, name, "LightsBaker.hlsl"): w(to_cpp(body) + "\n")
    w(to_cpp(extract_struct(ltext, "LocalReservoir", "LightsBaker.hlsl")) + "\n")
    w("groupshared LocalReservoir g_tile[32][32];\n")
    for name in ("ProcessFeedbackHistoryPreFilter", "ProcessFeedbackHistoryP0", "SampleLightGlobal", "MirrorCoord", "LSB_Address", "SampleLightLocalHistoric", "ConvertMotionVectorToPixelSpace", "Reproject",
                 "ProcessFeedbackHistoryP1a", "ProcessFeedbackHistoryP1b", "FillTile", "ProcessFeedbackHistoryP2", "InsertOneBit"):
        for body in extract_function(ltext, name, "LightsBaker.hlsl"): w(to_cpp(body) + "\n")
    w("groupshared uint g_localData[RTXPT_LIGHTING_LOCAL_PROXY_COUNT];\ngroupshared uint g_localDataRangeLR[RTXPT_LIGHTING_LOCAL_PROXY_COUNT];\n")
    for name in ("LastScanAndWriteOut", "ProcessFeedbackHistoryP3", "ClearFeedbackHistory", "ComputeProxyCounts"):
        for body in extract_function(ltext, name, "LightsBaker.hlsl"): w(to_cpp(body) + "\n")
    w("} // namespace lbfb\n")
    # PathTracerSample.hlsl: what the raygen shader does between two rays of the stable-plane passes (postProcessHit; FirstHitFromVBuffer of the fill pass)
    spath = os.path.join(ref, "Rtxpt/Shaders/PathTracerSample.hlsl")
    stext = strip_comments(open(spath, encoding="latin-1").read())
    w("// ======== PathTracerSample.hlsl (selected items)\n")
    for body in extract_function(stext, "postProcessHit", "PathTracerSample.hlsl"): w(to_cpp(body) + "\n")
    w("#if PATH_TRACER_MODE==PATH_TRACER_MODE_FILL_STABLE_PLANES\n")
    for body in extract_function(stext, "FirstHitFromVBuffer", "PathTracerSample.hlsl"): w(to_cpp(body) + "\n")
    w("#endif\n")
    # DenoisingGuidesBaker.hlsl: the specular-hit-distance fill-in that follows the noisy passes of a realtime frame
    dpath = os.path.join(ref, "Rtxpt/ProcessingPasses/DenoisingGuidesBaker.hlsl")
    dtext = strip_comments(open(dpath, encoding="latin-1").read())
    w("// ======== DenoisingGuidesBaker.hlsl (selected items)\nnamespace dgb {\n")
    w(to_cpp(extract_struct(dtext, "DenoisingGuidesBakerConstants", "DenoisingGuidesBaker.hlsl")) + "\n")
    w("template <class T> struct RWTexture2D { T* p = nullptr; uint w = 0, h = 0; T& operator[](int2 c) { return p[(size_t)c.y * w + c.x]; } void GetDimensions(uint& ow, uint& oh) const { ow = w; oh = h; } };      // whole-frame planes\n"
      "static RWTexture2D<float> u_Depth, u_SpecularHitT, u_ScratchFloat1; static DenoisingGuidesBakerConstants g_denoisingConstants;\n#define MAIN_BUFFER u_SpecularHitT\n#define SCRATCH_BUFFER u_ScratchFloat1\n")
    for name in ("SpecHitTNeighbourhood", "DenoiseSpecHitT"):
        for body in extract_function(dtext, name, "DenoisingGuidesBaker.hlsl"):      # (int2 >= uint2: HLSL converts the signed operand)
            w(to_cpp("\n".join(l for l in body.split("\n") if not re.match(r"\s*#\s*define", l))).replace("any(pixelPos >= uint2(", "any(uint2(pixelPos) >= uint2(") + "\n")
    w("} // namespace dgb\n")
    w("} // namespace hl\n")
    w(open(os.path.join(HERE, "hlsl_pt_wrappers.inc")).read())


def main_materials(ref):
    """third translation unit: plain C++ of the reference (Rtxpt/Materials/MaterialsBaker.{h,cpp}, PathTracer/Materials/MaterialPT.h) over mat_stubs.h"""
    w = sys.stdout.write
    w('// generated by oracle/refpin/hlsl_tu.py --materials -- never written to disk\n#include "%s/mat_stubs.h"\n' % HERE)
    w("using namespace hl;\n")
    mp = strip_comments(open(os.path.join(ref, "Rtxpt/Shaders/PathTracer/Materials/MaterialPT.h"), encoding="latin-1").read())
    w("\n".join(l for l in mp.split("\n") if not re.match(r"\s*#\s*include", l)) + "\n")
    hraw = open(os.path.join(ref, "Rtxpt/Materials/MaterialsBaker.h"), encoding="latin-1").read(); h = strip_comments(hraw)
    w(extract_struct(h, "PTTexture", "MaterialsBaker.h") + "\n")
    w(extract_struct(h, "PTMaterial", "MaterialsBaker.h") + "\n")
    c = strip_comments(open(os.path.join(ref, "Rtxpt/Materials/MaterialsBaker.cpp"), encoding="latin-1").read())
    for name in ("GetBindlessTextureIndex", "PTMaterial::IsEmissive", "PTMaterial::FillData", "PTMaterial::Read"):
        for body in extract_function(c, name, "MaterialsBaker.cpp"): w(body + "\n")
    # LightsBaker.cpp: the host-side light conversion (ConvertLight and the helpers it calls), over Donut light stand-ins
    pl = strip_comments(open(os.path.join(ref, SHADERS, "Lighting/PolymorphicLight.h"), encoding="latin-1").read())
    w("\n".join(l for l in pl.split("\n") if not re.match(r"\s*#\s*include", l)) + "\n")
    w(open(os.path.join(HERE, "light_stubs.inc")).read())
    lb = strip_comments(open(os.path.join(ref, "Rtxpt/Lighting/LightsBaker.cpp"), encoding="latin-1").read())
    for name in ("floatToUInt", "FLOAT3_to_R8G8B8_UNORM", "packLightColor", "OctWrap", "Encode_Oct", "NDirToOctUnorm32", "fp32ToFp16", "ConvertLight"):
        for body in extract_function(lb, name, "LightsBaker.cpp"): w(body + "\n")
    # LightsBaker::UpdateFrustumConsts: the plane extraction and normalisation (its text between the declaration of frustPlanes and the far plane), over Donut vector stand-ins
    lbraw = open(os.path.join(ref, "Rtxpt/Lighting/LightsBaker.cpp"), encoding="latin-1").read()
    w("namespace frustumpin {\nstruct float3 { float x, y, z; };\nstruct float4 { float x, y, z, w; float4() : x(0), y(0), z(0), w(0) {} float4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}\n"
      "    float3 xyz() const { return float3{x, y, z}; } float4 operator*(float s) const { return float4(x * s, y * s, z * s, w * s); } };\n"
      "static inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }\n"
      "struct Column { const float* m; int c; float operator[](int row) const { return m[4 * row + c]; } };\nstruct Matrix { const float* m; Column col(int c) const { return Column{m, c}; } };\nstruct Settings { Matrix ViewProjMatrix; };\n"
      "static void planes(const float* m16, float* out20) {\n    Settings settings{Matrix{m16}};\n")
    w(extract_range(lb, r"float4 frustPlanes\[6\];..// compute far plane", "LightsBaker.cpp", lbraw) + "\n")
    w("    for (int i = 0; i < 5; i++) { out20[4 * i] = frustPlanes[i].x; out20[4 * i + 1] = frustPlanes[i].y; out20[4 * i + 2] = frustPlanes[i].z; out20[4 * i + 3] = frustPlanes[i].w; }\n}\n} // namespace frustumpin\n"
      'extern "C" void reflight_frustum_planes(const float* m16, float* out20) { frustumpin::planes(m16, out20); }\n')
    w(open(os.path.join(HERE, "mat_wrappers.inc")).read())
    # ToneMapper host side: ColorUtils.h whole + the two ToneMappingPass members that build the colour transform, over Donut math stand-ins
    w(open(os.path.join(HERE, "color_stubs.inc")).read())
    cu = strip_comments(open(os.path.join(ref, "Rtxpt/ToneMapper/ColorUtils.h"), encoding="latin-1").read())
    w("\n".join(l for l in cu.split("\n") if not re.match(r"\s*#\s*(include|pragma)", l)) + "\n")
    cb = strip_comments(open(os.path.join(ref, "Rtxpt/ToneMapper/ToneMapping_cb.h"), encoding="latin-1").read())
    th = strip_comments(open(os.path.join(ref, "Rtxpt/ToneMapper/ToneMappingPasses.h"), encoding="latin-1").read())
    w("\n".join(l for l in cb.split("\n") if re.match(r"\s*#\s*define\s+TONEMAPPING_(AUTOEXPOSURE_CPU|EXPOSURE_KEY)\b", l)) + "\n")
    for text, name, path in ((cb, "ToneMapperOperator", "ToneMapping_cb.h"), (cb, "ToneMappingConstants", "ToneMapping_cb.h"), (th, "ExposureMode", "ToneMappingPasses.h"),
                             (th, "ToneMappingParameters", "ToneMappingPasses.h")):
        w(extract_struct(text, name, path) + "\n")
    w("COLORPIN_TONEMAPPINGPASS\n")
    traw = open(os.path.join(ref, "Rtxpt/ToneMapper/ToneMappingPasses.cpp"), encoding="latin-1").read(); tp = strip_comments(traw)
    for name in ("ToneMappingPass::SetParameters", "ToneMappingPass::UpdateExposureValue", "ToneMappingPass::UpdateWhiteBalanceTransform", "ToneMappingPass::UpdateColorTransform",
                 "ToneMappingPass::PreRender"):
        for body in extract_function(tp, name, "ToneMappingPasses.cpp"): w(body + "\n")
    # the constant-buffer fill inside ToneMappingPass::Render, as a member function of the stand-in
    w("ToneMappingConstants ToneMappingPass::FillConstants(uint viewIndex, bool enabled) {\n")
    w(extract_range(tp, r"ToneMappingConstants toneMappingConsts = \{\};..commandList->writeBuffer\(m_ToneMappingCB", "ToneMappingPasses.cpp", traw) + "\n")
    w("    return toneMappingConsts;\n}\n")
    # scene leaves: ExtendedScene.{h,cpp} classes + Load functions + CreateLeaf + FindEnvironmentLight, and the ranges of Sample.cpp that consume them
    w(open(os.path.join(HERE, "scene_stubs.inc")).read())
    eh = strip_comments(open(os.path.join(ref, "Rtxpt/SampleCommon/ExtendedScene.h"), encoding="latin-1").read()).replace("[[nodiscard]]", "")
    for name in ("LightSamplerLink", "LightExtension", "SpotLightEx", "PointLightEx", "EnvironmentLight", "PerspectiveCameraEx", "SampleSettings"):
        w(extract_struct(eh, name, "ExtendedScene.h") + "\n")
    w("SCENEPIN_EXTRA_LEAVES\n")
    ecraw = open(os.path.join(ref, "Rtxpt/SampleCommon/ExtendedScene.cpp"), encoding="latin-1").read(); ec = strip_comments(ecraw)
    for name in ("LightExtension::Load", "SpotLightEx::Load", "PointLightEx::Load", "EnvironmentLight::Load", "PerspectiveCameraEx::Load", "SampleSettings::Load"):
        for body in extract_function(ec, name, "ExtendedScene.cpp"): w(body + "\n")
    w(extract_range(ec, r"ExtendedSceneTypeFactory::CreateLeaf\(..ExtendedSceneTypeFactory::CreateMesh\(\)", "ExtendedScene.cpp", ecraw) + "\n")
    w(extract_range(ec, r"^std::shared_ptr<EnvironmentLight> FindEnvironmentLight..^void EnvironmentLight::FillLightConstants", "ExtendedScene.cpp", ecraw) + "\n")
    sc = strip_comments(open(os.path.join(ref, "Rtxpt/SampleCommon/SampleCommon.cpp"), encoding="latin-1").read())
    w(extract_range(sc, r"^std::vector<std::string> JsonLoadStringVector..^uint64_t GetEstimatedTextureSize", "SampleCommon.cpp") + "\n")
    w(open(os.path.join(HERE, "scene_wrappers.inc")).read())
    smraw = open(os.path.join(ref, "Rtxpt/Sample.cpp"), encoding="latin-1").read(); sm = strip_comments(smraw)
    for body in extract_function(sm, "Sample::UpdateCameraFromScene", "Sample.cpp"): w(body + "\n")
    w("void Sample::SensibleDefaults() {\n" + extract_range(sm, r"m_ui\.ToneMappingParams\.exposureCompensation = 2\.0f;..std::shared_ptr<EnvironmentLight> envLight = FindEnvironmentLight", "Sample.cpp", smraw) + "\n}\n")
    w("void Sample::CleanUpLights() {\n" + extract_range(sm, r"for \(int i = \(int\)m_lights\.size\(\)..if\( m_envMapLocalPath != \"\" \)", "Sample.cpp", smraw) + "\n}\n")
    w("void Sample::ApplySettings() {\n" + extract_range(sm, r"std::shared_ptr<SampleSettings> settings = m_scene->GetSampleSettingsNode\(\);..if \(m_cmdLine\.stopAnimations\)", "Sample.cpp", smraw) + "\n}\n")
    w(open(os.path.join(HERE, "scene_driver.inc")).read())
    w(open(os.path.join(HERE, "color_wrappers.inc")).read())


def main():
    if "--materials" in sys.argv:
        sys.argv.remove("--materials"); return main_materials(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    if "--integrator" in sys.argv:
        sys.argv.remove("--integrator"); return main_pt(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    w = sys.stdout.write
    w('// generated by oracle/refpin/hlsl_tu.py -- never written to disk\n#include "%s/hlsl_shim.h"\nnamespace hl {\n' % HERE)
    for rel, names in PLAN:
        path = os.path.join(ref, rel[1:]) if rel.startswith("@") else os.path.join(ref, SHADERS, rel)      # "@": relative to the reference root
        raw = open(path, encoding="latin-1").read()
        text = strip_comments(raw)
        for name in names:
            if name.startswith("#"):
                for m in re.finditer(r"^[ \t]*#define[ \t]+(" + re.escape(name[1:]) + r"\w*)[ \t]+(\S+)", text, re.M):
                    w("static const float %s = %s;\n" % (m.group(1), m.group(2)))
                continue
            if name.startswith("text "): w(name[5:] + "\n"); continue
            if name == "pp":
                w("\n".join(l for l in text.split("\n") if re.match(r"\s*#", l) and not re.match(r"\s*#\s*include", l)) + "\n"); continue
            if name.startswith("struct "): w("// ---- %s : %s\n" % (rel, name)); w(to_cpp(extract_struct(text, name[7:], rel))); w("\n"); continue
            if name.startswith("range "): w("// ---- %s : %s\n" % (rel, name)); w(to_cpp(extract_range(text, name[6:], rel, raw))); w("\n"); continue
            for body in extract_function(text, name, rel):
                w("// ---- %s : %s\n" % (rel, name)); w(to_cpp(body)); w("\n")
    w("} // namespace hl\n")
    w(open(os.path.join(HERE, "hlsl_wrappers.inc")).read())


if __name__ == "__main__":
    main()
