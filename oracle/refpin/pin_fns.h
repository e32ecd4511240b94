// ORACLE pin (test infrastructure only): ids and arities of the functions that exist twice — as reference text compiled through
// hlsl_shim.h (refhlsl_call, librefpin_hlsl.so) and as the oracle's restatement (ptref_pin_call, libptref.so). tests/test_oracle_refpin_hlsl.py
// feeds both the same inputs and compares the outputs bit for bit.
#pragma once
enum PinFn {
    PIN_evalFresnelSchlick = 0,          // (f0, f90, cosTheta) -> F                           Fresnel.hlsli:30-33
    PIN_evalFresnelSchlick3,             // (f0.rgb, f90, cosTheta) -> F.rgb                   Fresnel.hlsli:25-28
    PIN_evalFresnelDielectric,           // (eta, cosThetaI) -> (F, cosThetaT)                 Fresnel.hlsli:52-77
    PIN_evalNdfGGX,                      // (alpha, cosTheta) -> D                             Microfacet.hlsli:33-38
    PIN_evalPdfGGX_BVNDF,                // (alpha, i.xyz, m.xyz) -> pdf                       Microfacet.hlsli:105-128
    PIN_sampleGGX_BVNDF,                 // (alpha, i.xyz, u.xy) -> m.xyz                      Microfacet.hlsli:185-207
    PIN_evalLambdaGGX,                   // (alphaSqr, cosTheta) -> lambda                     Microfacet.hlsli:232-239
    PIN_evalMaskingSmithGGXCorrelated,   // (alpha, cosThetaI, cosThetaO) -> G                 Microfacet.hlsli:267-273
    PIN_ndir_to_oct_equal_area_unorm,    // (n.xyz) -> p.xy                                    MathHelpers.hlsli:185-201
    PIN_oct_to_ndir_equal_area_unorm,    // (p.xy) -> n.xyz                                    MathHelpers.hlsli:207-226
    PIN_sample_disk,                     // (u.xy) -> p.xy                                     MathHelpers.hlsli:238-246
    PIN_sample_disk_concentric,          // (u.xy) -> p.xy                                     MathHelpers.hlsli:288-304
    PIN_sample_cosine_hemisphere_concentric, // (u.xy) -> (d.xyz, pdf)                         MathHelpers.hlsli:311-317
    PIN_perp_stark,                      // (u.xyz) -> v.xyz                                   MathHelpers.hlsli:436-448
    PIN_ComputeRayOrigin,                // (p.xyz, n.xyz) -> o.xyz                            PathTracerHelpers.hlsli:29-42
    PIN_FastSqrt,                        // (x) -> y                                           Utils.hlsli:484-487
    PIN_FastACos,                        // (x) -> y                                           Utils.hlsli:489-497
    PIN_ComputeRayConeSpreadAngleExpansionByScatterPDF, // (pdf, growthFactor) -> angle        PathTracerHelpers.hlsli:189-192
    PIN_ComputeNewScatterFireflyFilterK, // (currentK, bouncePDF, lobeP) -> K                  PathTracerHelpers.hlsli:195-203
    PIN_FireflyFilter,                   // (signal.rgb, threshold, K) -> signal.rgb           PathTracerHelpers.hlsli:206-213
    PIN_FireflyFilterShort,              // (signalAverage, threshold, K) -> scale             PathTracerHelpers.hlsli:214-219
    PIN_ComputeLowGrazingAngleFalloff,   // (l.xyz, n.xyz, from, range) -> falloff             PathTracerHelpers.hlsli:48-52
    PIN_COUNT
};
static const int kPinArity[PIN_COUNT][2] = {   // {inputs, outputs} in floats
    {3, 1}, {5, 3}, {2, 2}, {2, 1}, {7, 1}, {6, 3}, {2, 1}, {3, 1}, {3, 2}, {2, 3}, {2, 2}, {2, 2}, {2, 4}, {3, 3}, {6, 3}, {1, 1}, {1, 1}, {2, 1}, {3, 1}, {5, 3}, {3, 1}, {8, 1},
};
