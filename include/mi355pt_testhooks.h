/* mi355pt — evaluation hooks for the known-answer tests. NOT part of the product's ABI: the shipped rtxpt_amd/libmi355pt.so does not export anything declared here.
 * `make -C rtxpt_amd/csrc` also builds rtxpt_amd/libmi355pt_testhooks.so — the same sources with -DMI355PT_TEST_HOOKS — which exports everything include/mi355pt.h
 * declares plus the entry point below; tests/ load that variant where they need it (rtxpt_amd.PathTracer(test_hooks=True)). */
#ifndef MI355PT_TESTHOOKS_H
#define MI355PT_TESTHOOKS_H
#include "mi355pt.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Device-side evaluation of the product's own leaf functions, one thread per row, so that a test can compare them bit for bit with the outputs of the reference's text
 * (tests/golden/refpin_hlsl_golden.npz) or of the oracle. kind (rtxpt_amd/csrc/pt_wavefront.hip k_probe): 0 deterministic math (fn, x, y); 1 binary16 round trip;
 * 2 sample streams (pixel, vertex, sample, seed, generator, count); 3 whole-BSDF eval / sample / pdf; 4 camera rays; 5 leaf functions pinned to the reference text
 * (Fresnel, microfacet, octahedral maps, disk / hemisphere sampling, ComputeRayOrigin, firefly filter ...); 6 polymorphic lights; 7 the half-typed operators of the lp16 build;
 * 8 Bridge::loadSurface (45 words per hit); 9 EnvMap::EvalLocal on the baked cube; 10 the traversal's alpha test. `n` rows in, `n` rows out; the row layouts are the probe's. */
int32_t pt_probe(pt_context* ctx, int32_t kind, const void* in, size_t inBytes, void* out, size_t outBytes, uint32_t n);
#ifdef __cplusplus
}
#endif
#endif
