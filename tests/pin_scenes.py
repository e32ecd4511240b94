"""Cases shared by tests/golden/make_reference_integrator_golden.py, tests/test_oracle_refpin_integrator.py and the GPU fixture test: name ->
(scene factory, settings, width, height, first sample, sample count). Small frames: the point is which code runs, not how many pixels."""
from rtxpt_amd import scenes


def cases():
    c2 = lambda: scenes.cornell_box("C2")
    return {
        "c1": (lambda: scenes.cornell_box("C1"), scenes.config_settings("C1"), 48, 48, 0, 2),                      # Lambert, 2 bounces, no RR
        "c2": (c2, scenes.config_settings("C2"), 64, 36, 0, 2),                                                    # StandardBSDF mix, env, nested glass, RR
        "c2_firefly": (c2, scenes.default_settings(fireflyFilterThreshold=2.5), 64, 36, 3, 2),
        "c2_nee3": (c2, scenes.default_settings(NEEFullSamples=3), 64, 36, 0, 2),                                  # HandleNEE_MultipleSamples
        "c2_nee_off": (c2, scenes.default_settings(NEEEnabled=0), 64, 36, 0, 2),
        "c2_nested2_norr_nold": (c2, scenes.default_settings(nestedDielectricsQuality=2, enableRussianRoulette=0, enableLDSamplerForBSDF=0), 64, 36, 0, 2),
        "c2_nested0_uniform": (c2, scenes.default_settings(nestedDielectricsQuality=0, NEEType=0), 64, 36, 0, 2),
        "bistro_like": (lambda: scenes.bistro_like(scale=0.02, tex_size=128), scenes.default_settings(), 96, 54, 0, 2),      # alpha test, textures, normal maps, emissive triangles, env quads
        "bistro_like_c5": (lambda: scenes.bistro_like(scale=0.01, tex_size=64, animated=True), scenes.default_settings(), 96, 54, 0, 2),   # + nested-dielectric props
    }
