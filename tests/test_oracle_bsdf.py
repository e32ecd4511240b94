"""Oracle self-consistency of the StandardBSDF restatement (SURVEY.md §7 gate 1b): pdf integrates to 1, sample weight equals
eval/pdf for non-delta lobes, white furnace <= 1, every lobe mix. These properties pin the oracle independently of the GPU."""
import ctypes

import numpy as np
import pytest

from oracle import ptref

L = ptref.lib()


def probe(params, wi, w, mode, thin=1, model=2):
    p = np.ascontiguousarray(params, np.float32); wi = np.ascontiguousarray(wi, np.float32); w = np.ascontiguousarray(w, np.float32)
    out = np.zeros(10, np.float32)
    L.ptref_bsdf_probe(p.ctypes.data_as(ctypes.c_void_p), thin, model, wi.ctypes.data_as(ctypes.c_void_p), w.ctypes.data_as(ctypes.c_void_p), mode, out.ctypes.data_as(ctypes.c_void_p))
    return out


def mat(diffuse=(0.8, 0.8, 0.8), F0=0.04, rough=0.5, metallic=0.0, trans=(1, 1, 1), dtrans=0.0, strans=0.0, eta=1 / 1.5):
    return list(diffuse) + [F0] * 3 + [rough, metallic] + list(trans) + [dtrans, strans, eta]


MATERIALS = {
    "lambert_like": (mat(F0=0.0, rough=1.0), 1, 0),
    "frostbite_rough": (mat(rough=0.7), 1, 2),
    "glossy_dielectric": (mat(rough=0.3), 1, 2),
    "metal": (mat(diffuse=(0, 0, 0), F0=0.9, rough=0.25, metallic=1.0), 1, 2),
    "diffuse_transmission": (mat(rough=0.6, dtrans=0.5), 1, 2),
    "rough_glass": (mat(diffuse=(0, 0, 0), rough=0.4, strans=1.0, trans=(0.9, 0.9, 0.9)), 0, 2),
    "mixed": (mat(rough=0.5, strans=0.4, dtrans=0.3), 0, 2),
}


def sphere_dirs(n):
    k = np.arange(n) + 0.5
    z = 1 - 2 * k / n
    phi = k * np.pi * (3 - np.sqrt(5))
    r = np.sqrt(1 - z * z)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)


@pytest.mark.parametrize("name", list(MATERIALS))
def test_pdf_integrates_to_one_and_furnace(name):
    params, thin, model = MATERIALS[name]
    wi = np.array([0.5, 0.1, np.sqrt(1 - 0.26)], np.float32)
    n = 40000
    dirs = sphere_dirs(n)
    pdf = np.zeros(n); f = np.zeros((n, 3))
    for i in range(n):
        o = probe(params, wi, dirs[i], 0, thin, model)
        pdf[i] = o[4]; f[i] = o[:3]
    dw = 4 * np.pi / n
    total_pdf = pdf.sum() * dw
    assert abs(total_pdf - 1.0) < 0.03, (name, total_pdf)
    albedo = f.sum(0) * dw          # eval already contains the cosine
    assert np.all(albedo <= 1.02), (name, albedo)


@pytest.mark.parametrize("name", list(MATERIALS))
def test_sample_weight_equals_eval_over_pdf(name):
    params, thin, model = MATERIALS[name]
    wi = np.array([0.3, -0.2, np.sqrt(1 - 0.13)], np.float32)
    rng = np.random.default_rng(5)
    checked = 0
    for _ in range(400):
        u = rng.random(3).astype(np.float32)
        s = probe(params, wi, u, 1, thin, model)
        if s[9] == 0 or s[3] == 0:
            continue
        wo = s[:3]
        e = probe(params, wi, wo, 0, thin, model)
        assert abs(np.linalg.norm(wo) - 1) < 1e-4
        assert abs(e[4] - s[3]) <= 2e-3 * max(1.0, s[3]), (name, e[4], s[3])           # evalPdf(wo) == sampled pdf
        # weight == eval / pdf only for the lobe that was sampled when lobes do not overlap; for overlapping lobes the reference
        # returns f_lobe / (p_lobe * pdf_lobe), so we check the unbiasedness identity through the expectation below instead
        checked += 1
    assert checked > 50


@pytest.mark.parametrize("name", ["lambert_like", "frostbite_rough", "glossy_dielectric", "metal"])
def test_sampling_estimates_albedo(name):
    """E[weight] over sample() equals the integral of eval() (importance sampling is unbiased), reflection-only materials."""
    params, thin, model = MATERIALS[name]
    wi = np.array([0.0, 0.4, np.sqrt(1 - 0.16)], np.float32)
    rng = np.random.default_rng(9)
    n = 20000
    acc = np.zeros(3)
    for _ in range(n):
        s = probe(params, wi, rng.random(3).astype(np.float32), 1, thin, model)
        if s[9] != 0:
            acc += s[4:7]
    est = acc / n
    dirs = sphere_dirs(40000)
    f = np.array([probe(params, wi, d, 0, thin, model)[:3] for d in dirs]).sum(0) * (4 * np.pi / 40000)
    assert np.allclose(est, f, rtol=0.05, atol=0.01), (name, est, f)


def test_delta_lobes():
    params = mat(diffuse=(0, 0, 0), rough=0.0, strans=1.0)
    wi = np.array([0.0, 0.6, 0.8], np.float32)
    refl = trans = 0
    for u in np.linspace(0.01, 0.99, 50):
        s = probe(params, wi, np.array([0.3, 0.3, u], np.float32), 1, thin=0)
        assert s[9] == 1 and s[3] == 0           # valid, pdf == 0 for delta events (BxDF.hlsli:956-957)
        lobe = int(s[7])
        if lobe == 0x04:
            refl += 1; assert np.allclose(s[:3], [-wi[0], -wi[1], wi[2]], atol=1e-6)
        else:
            assert lobe == 0x40; trans += 1; assert s[2] < 0
    assert refl > 0 and trans > refl             # Fresnel at 37 deg, eta 1/1.5: mostly transmission
