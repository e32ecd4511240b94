"""Seeded input rows for the functions pinned against reference text (oracle/refpin/pin_fns.h): shared by the golden generator
(tests/golden/make_refpin_hlsl_golden.py) and tests/test_oracle_refpin_hlsl.py. Each generator covers the function's domain plus its edges."""
import numpy as np


def _unit(rng, n, upper=False):
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    if upper: v[:, 2] = np.abs(v[:, 2])
    return v


def _axes():
    return np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [0.70710678, 0.70710678, 0], [0.57735027, 0.57735027, 0.57735027],
                     [0, 0.70710678, -0.70710678]], np.float64)


def _logu(rng, n, lo, hi):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), n))


def rows(name, n, seed):
    rng = np.random.default_rng(seed)
    u2 = lambda: np.vstack([rng.uniform(0, 1, (n - 6, 2)), [[0, 0], [0.5, 0.5], [0.999999, 0.999999], [0.25, 0.75], [0.5, 0.1], [0, 0.5]]])
    alpha = lambda k: np.concatenate([rng.uniform(0.0064, 1.0, k - 2), [0.0064, 1.0]])
    if name == "evalFresnelSchlick": a = np.column_stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.uniform(-0.1, 1.1, n)])
    elif name == "evalFresnelSchlick3": a = np.column_stack([rng.uniform(0, 1, (n, 3)), rng.uniform(0, 1, n), rng.uniform(-0.1, 1.1, n)])
    elif name == "evalFresnelDielectric": a = np.column_stack([rng.uniform(0.4, 2.5, n), rng.uniform(-1, 1, n)])
    elif name == "evalNdfGGX": a = np.column_stack([alpha(n), rng.uniform(0, 1, n)])
    elif name == "evalPdfGGX_BVNDF": a = np.column_stack([alpha(n), _unit(rng, n, True), _unit(rng, n, True)])
    elif name == "sampleGGX_BVNDF": a = np.column_stack([alpha(n), _unit(rng, n, True), u2()])
    elif name == "evalLambdaGGX": a = np.column_stack([rng.uniform(0, 1, n), rng.uniform(-0.2, 1, n)])
    elif name == "evalMaskingSmithGGXCorrelated": a = np.column_stack([alpha(n), rng.uniform(-0.1, 1, n), rng.uniform(-0.1, 1, n)])
    elif name in ("ndir_to_oct_equal_area_unorm", "perp_stark"): a = np.vstack([_unit(rng, n - 9), _axes()])
    elif name == "oct_to_ndir_equal_area_unorm": a = np.vstack([rng.uniform(0, 1, (n - 6, 2)), [[0, 0], [1, 1], [0.5, 0.5], [0, 1], [0.5, 0], [1, 0.25]]])
    elif name in ("sample_disk", "sample_disk_concentric", "sample_cosine_hemisphere_concentric"): a = u2()
    elif name == "ComputeRayOrigin":
        p = _logu(rng, (n, 3), 1e-4, 1e3) * rng.choice([-1.0, 1.0], (n, 3)); p[:4] = [[0, 0, 0], [0.0625, -0.0625, 1], [1e-9, 5, -5], [0.06, 0.07, -0.06]]
        a = np.column_stack([p, _unit(rng, n)])
    elif name == "FastSqrt": a = np.concatenate([rng.uniform(0, 4, n - 3), [0, 1, 4]])[:, None]
    elif name == "FastACos": a = np.concatenate([rng.uniform(-1, 1, n - 3), [-1, 0, 1]])[:, None]
    elif name == "ComputeRayConeSpreadAngleExpansionByScatterPDF": a = np.column_stack([_logu(rng, n, 1e-3, 1e4), rng.choice([0.3, 1.0], n)])
    elif name == "ComputeNewScatterFireflyFilterK":
        pdf = _logu(rng, n, 1e-3, 1e4); pdf[:3] = 0
        a = np.column_stack([_logu(rng, n, 1e-5, 1), pdf, rng.uniform(0, 1, n)])
    elif name == "FireflyFilter": a = np.column_stack([_logu(rng, (n, 3), 1e-3, 1e4), _logu(rng, n, 0.1, 100), _logu(rng, n, 1e-5, 1)])
    elif name == "FireflyFilterShort": a = np.column_stack([_logu(rng, n, 1e-3, 1e4), _logu(rng, n, 0.1, 100), _logu(rng, n, 1e-5, 1)])
    elif name == "ComputeLowGrazingAngleFalloff": a = np.column_stack([_unit(rng, n), _unit(rng, n), rng.uniform(0, 0.5, n), rng.uniform(0.01, 1, n)])
    else: raise KeyError(name)
    return np.ascontiguousarray(a, dtype=np.float32)


def bsdf_cases(n, seed):
    """Rows for the whole-BSDF pin: [14 material params, thin, diffuseModel, wi.xyz, w.xyz (direction for eval / random numbers for sample), mode].
    Materials span every lobe mix: metals, dielectrics, delta (roughness below the GGX clamp), diffuse / specular transmission, eta on both sides of 1."""
    rng = np.random.default_rng(seed)
    R = np.zeros((n, 23), np.float32)
    for i in range(n):
        kind = rng.integers(0, 8)
        diffuse = rng.uniform(0, 1, 3); spec = rng.uniform(0, 1, 3) if kind in (1, 5) else np.full(3, rng.choice([0.0, 0.04, 0.08]))
        rough = rng.choice([0.0, 0.05, 0.079, 0.081, 0.3, 0.7, 1.0]) if rng.random() < 0.4 else rng.uniform(0, 1)
        metallic = [0, 1, 0, rng.uniform(0, 1), 0, 1, 0, rng.uniform(0, 1)][kind]
        trans = rng.uniform(0, 1, 3)
        dtrans = [0, 0, rng.uniform(0, 1), 0, 1, 0, rng.uniform(0, 1), 0][kind]
        strans = [0, 0, 0, rng.uniform(0, 1), 0, 0, rng.uniform(0, 1), 1][kind]
        eta = rng.choice([1.0, 1 / 1.5, 1.5, 1 / 1.33, 1.33]) if rng.random() < 0.7 else rng.uniform(0.5, 2.0)
        if kind == 0 and rng.random() < 0.3: diffuse[:] = 0
        wi = _unit(rng, 1, upper=rng.random() < 0.85)[0]
        mode = int(rng.integers(0, 2))
        w = _unit(rng, 1)[0] if mode == 0 else rng.uniform(0, 1, 3)
        if mode == 1 and rng.random() < 0.05: w[2] = rng.choice([0.0, 0.999999])
        R[i, :14] = list(diffuse) + list(spec) + [rough, metallic] + list(trans) + [dtrans, strans, eta]
        R[i, 14] = rng.integers(0, 2); R[i, 15] = rng.choice([0, 2]); R[i, 16:19] = wi; R[i, 19:22] = w; R[i, 22] = mode
    return R


def stream_cases(n, seed):
    """uint32 rows [packedPixel, vertexIndex, sampleIndex, effectSeed, kind, count] for the stateless sample generators (kinds: see ptref_sample_stream)."""
    rng = np.random.default_rng(seed)
    C = np.zeros((n, 6), np.uint32)
    C[:, 0] = (rng.integers(0, 3840, n) << 16) | rng.integers(0, 2160, n)
    C[:, 1] = rng.integers(0, 40, n); C[:, 2] = np.where(rng.random(n) < 0.2, rng.integers(0, 2**32, n, dtype=np.uint64), rng.integers(0, 4096, n)).astype(np.uint32)
    C[:, 3] = rng.choice([0, 1, 2, 3, 5, 6], n); C[:, 4] = rng.integers(0, 5, n)
    C[:, 5] = np.where(C[:, 4] < 2, rng.integers(1, 5, n), rng.integers(1, 9, n))
    return C


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _f16(x):
    return np.asarray(x, np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)      # round-to-nearest-even, as the oracle's f32tof16


def light_inputs(n, seed, probe):
    """Input rows (uint32 words) for the five light probes. `probe(kind, words)` evaluates a probe (used to obtain packed colours, triangle records and
    oct-encoded axes for the records of kind 2 / 3, so the test data depends only on the implementation under test being self-consistent)."""
    rng = np.random.default_rng(seed)
    rad = _logu(rng, (n, 3), 1e-3, 1e5).astype(np.float32); rad[:3] = [[0, 0, 0], [1, 1, 1], [0, 5, 0]]; rad[3:8, rng.integers(0, 3)] = 0
    k0 = _bits(rad)
    base = rng.uniform(-50, 50, (n, 3)); e1 = rng.normal(size=(n, 3)) * _logu(rng, (n, 1), 1e-2, 5); e2 = rng.normal(size=(n, 3)) * _logu(rng, (n, 1), 1e-2, 5)
    e2[:2] = e1[:2] * 2                                                   # degenerate triangles
    k1 = _bits(np.column_stack([base, e1, e2, rad]))
    tri = probe(1, k1)
    col = probe(0, k0)
    viewer = rng.uniform(-60, 60, (n, 3)); u = np.vstack([rng.uniform(0, 1, (n - 3, 2)), [[0, 0], [0.999999, 0.999999], [0.5, 0.5]]])
    # sphere records: centre, radius (fp16), colour, optional spot shaping
    axis = probe(4, _bits(_unit(rng, n)))[:, 0]
    sph = np.zeros((n, 12), np.uint32)
    c = rng.uniform(-30, 30, (n, 3)); sph[:, 0:3] = _bits(c)
    spot = rng.random(n) < 0.5; minf = rng.random(n) < 0.3
    sph[:, 3] = col[:, 0] | (0 << 24) | np.where(spot, 1 << 28, 0).astype(np.uint32) | np.where(spot & minf, 1 << 30, 0).astype(np.uint32)
    sph[:, 6] = _f16(_logu(rng, n, 1e-2, 10)); sph[:, 7] = col[:, 1]
    sph[:, 9] = axis; sph[:, 10] = _f16(rng.uniform(-0.5, 0.99, n)) | (_f16(rng.uniform(0.0, 0.5, n)) << 16); sph[:, 11] = 11
    vs = viewer.copy(); vs[:4] = c[:4] + 1e-3                             # viewers inside the sphere
    # environment quads: node (x, y) of a dim x dim equal-area octahedral grid
    env = np.zeros((n, 12), np.uint32)
    dim = 2 ** rng.integers(0, 9, n); nx = (rng.random(n) * dim).astype(np.uint32); ny = (rng.random(n) * dim).astype(np.uint32)
    env[:, 3] = col[:, 0] | (5 << 24); env[:, 4] = (nx << 16) | ny; env[:, 5] = dim.astype(np.uint32) << 16; env[:, 6] = _bits(rng.uniform(0, 1, n)); env[:, 7] = col[:, 1]; env[:, 11] = 13
    recs = np.vstack([tri, sph, env]); rnd = np.vstack([u, u, u]); vw = np.vstack([viewer, vs, viewer])
    k2 = np.column_stack([recs, _bits(rnd), _bits(vw)])
    smp = probe(2, k2[:n])[:, 0:3]                                        # sample positions on the triangles
    k3 = np.column_stack([tri, _bits(viewer), smp])
    k4 = _bits(np.vstack([_unit(rng, n - 9), _axes()]))
    return {0: k0, 1: k1, 2: k2, 3: k3, 4: k4}


def tonemap_cases(seed, dtype):
    """(params record, rgba rows) per case: the six operators x {manual exposure, CPU auto exposure, disabled, unclamped} with a random colour transform."""
    rng = np.random.default_rng(seed)
    cases = []
    for op in range(6):
        for variant in range(4):
            p = np.zeros((), dtype=dtype)
            p["whiteScale"] = rng.uniform(2, 12); p["whiteMaxLuminance"] = rng.uniform(0.5, 4); p["toneMapOperator"] = op; p["clamped"] = 0 if variant == 3 else 1
            p["autoExposure"] = 1 if variant == 1 else 0; p["avgLuminance"] = rng.uniform(0.01, 2); p["autoExposureLumValueMin"] = 2.0 ** -4; p["autoExposureLumValueMax"] = 2.0 ** 4
            M = np.eye(3) * rng.uniform(0.2, 3) + rng.uniform(-0.05, 0.05, (3, 3)); p["colorTransform"] = M.reshape(-1)
            p["enabled"] = 0 if variant == 2 else 1
            rgba = np.column_stack([_logu(rng, (400, 3), 1e-4, 50), rng.uniform(0, 1, 400)]).astype(np.float32); rgba[:3, :3] = [[0, 0, 0], [1, 1, 1], [0.004, 0.003, 0.005]]
            cases.append((p, rgba))
    return cases


def lightbake_inputs(seed, light_records):
    """(kind-0 rows = the given light records, pyramid, kind-1 rows): a random importance pyramid (64^2 down to 1^2) and every kind of node query."""
    rng = np.random.default_rng(seed)
    pyr = [np.concatenate([_logu(rng, (d, d, 3), 1e-3, 1e3), _logu(rng, (d, d, 1), 1e-5, 1e4)], axis=2).astype(np.float32) for d in (64, 32, 16, 8, 4, 2, 1)]
    pyr[2][:2, :2, 3] = 0.0
    rows = []
    for k in range(3000):
        lg = int(rng.integers(0, 7)); dim = 1 << lg
        rows.append([dim, int(rng.integers(0, dim)), int(rng.integers(0, dim)), int(rng.integers(0, 4096)), int(rng.integers(0, 7))])
    return np.ascontiguousarray(light_records, np.uint32), pyr, np.array(rows, np.uint32)
