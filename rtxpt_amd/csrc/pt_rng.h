// mi355pt device/host leaf library — stateless sample generators
// Part of the PRODUCT path (libmi355pt.so). Written to the arithmetic contract stated in pt_vec.h so that the HIP kernels
// reproduce the reference estimator bit-for-bit against the independent CPU oracle used by the tests.
// Reference anchors are cited per function (paths relative to /root/reference/Rtxpt/Shaders/PathTracer/ unless noted).
// Restates, function by function:
//   Rtxpt/Shaders/PathTracer/Utils/NoiseAndSequences.hlsli:58-86 (Hash32, Hash32Combine, Hash32ToFloat),
//   :130-229 (bhos_sobol, bhos_owen_hash, bhos_owen_scramble),
//   Rtxpt/Shaders/PathTracer/Utils/StatelessSampleGenerators.hlsli:18-50 (SampleGeneratorVertexBase),
//   :62-171 (SampleSequenceGenerator), :179-232 (UniformSampleSequenceGenerator),
//   Rtxpt/Shaders/PathTracer/Utils/SampleGenerators.hlsli:16-52 (effect seeds, sampleNext1D).
// All of it is exact 32-bit integer arithmetic and must match the reference bit-for-bit; it is pinned against the
#pragma once
#include "pt_vec.h"

namespace ptk {
#pragma clang force_cuda_host_device begin

// NoiseAndSequences.hlsli:58-69
static inline uint Hash32(uint x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0xf35a2d97u; x ^= x >> 15;
    return x;
}
// :71-74
static inline uint Hash32Combine(uint seed, uint value) {
    return seed ^ (Hash32(value) + 0x9e3779b9u + (seed << 6) + (seed >> 2));
}
// :81-86
static inline float Hash32ToFloat(uint hash) { return (float)(hash >> 8) / 16777216.0f; }

// :135-180 direction numbers (5 dimensions x 32 bits)
static __device__ const uint kSobolDirections[5][32] = {
    {0x80000000u, 0x40000000u, 0x20000000u, 0x10000000u, 0x08000000u, 0x04000000u, 0x02000000u, 0x01000000u,
     0x00800000u, 0x00400000u, 0x00200000u, 0x00100000u, 0x00080000u, 0x00040000u, 0x00020000u, 0x00010000u,
     0x00008000u, 0x00004000u, 0x00002000u, 0x00001000u, 0x00000800u, 0x00000400u, 0x00000200u, 0x00000100u,
     0x00000080u, 0x00000040u, 0x00000020u, 0x00000010u, 0x00000008u, 0x00000004u, 0x00000002u, 0x00000001u},
    {0x80000000u, 0xc0000000u, 0xa0000000u, 0xf0000000u, 0x88000000u, 0xcc000000u, 0xaa000000u, 0xff000000u,
     0x80800000u, 0xc0c00000u, 0xa0a00000u, 0xf0f00000u, 0x88880000u, 0xcccc0000u, 0xaaaa0000u, 0xffff0000u,
     0x80008000u, 0xc000c000u, 0xa000a000u, 0xf000f000u, 0x88008800u, 0xcc00cc00u, 0xaa00aa00u, 0xff00ff00u,
     0x80808080u, 0xc0c0c0c0u, 0xa0a0a0a0u, 0xf0f0f0f0u, 0x88888888u, 0xccccccccu, 0xaaaaaaaau, 0xffffffffu},
    {0x80000000u, 0xc0000000u, 0x60000000u, 0x90000000u, 0xe8000000u, 0x5c000000u, 0x8e000000u, 0xc5000000u,
     0x68800000u, 0x9cc00000u, 0xee600000u, 0x55900000u, 0x80680000u, 0xc09c0000u, 0x60ee0000u, 0x90550000u,
     0xe8808000u, 0x5cc0c000u, 0x8e606000u, 0xc5909000u, 0x6868e800u, 0x9c9c5c00u, 0xeeee8e00u, 0x5555c500u,
     0x8000e880u, 0xc0005cc0u, 0x60008e60u, 0x9000c590u, 0xe8006868u, 0x5c009c9cu, 0x8e00eeeeu, 0xc5005555u},
    {0x80000000u, 0xc0000000u, 0x20000000u, 0x50000000u, 0xf8000000u, 0x74000000u, 0xa2000000u, 0x93000000u,
     0xd8800000u, 0x25400000u, 0x59e00000u, 0xe6d00000u, 0x78080000u, 0xb40c0000u, 0x82020000u, 0xc3050000u,
     0x208f8000u, 0x51474000u, 0xfbea2000u, 0x75d93000u, 0xa0858800u, 0x914e5400u, 0xdbe79e00u, 0x25db6d00u,
     0x58800080u, 0xe54000c0u, 0x79e00020u, 0xb6d00050u, 0x800800f8u, 0xc00c0074u, 0x200200a2u, 0x50050093u},
    {0x80000000u, 0x40000000u, 0x20000000u, 0xb0000000u, 0xf8000000u, 0xdc000000u, 0x7a000000u, 0x9d000000u,
     0x5a800000u, 0x2fc00000u, 0xa1600000u, 0xf0b00000u, 0xda880000u, 0x6fc40000u, 0x81620000u, 0x40bb0000u,
     0x22878000u, 0xb3c9c000u, 0xfb65a000u, 0xddb2d000u, 0x78022800u, 0x9c0b3c00u, 0x5a0fb600u, 0x2d0ddb00u,
     0xa2878080u, 0xf3c9c040u, 0xdb65a020u, 0x6db2d0b0u, 0x800228f8u, 0x400b3cdcu, 0x200fb67au, 0xb00ddb9du},
};
// :130-191
static inline uint bhos_sobol(uint index, uint dimension) {
    uint X = 0u;
    for (uint bit = 0; bit < 32; bit++) {
        uint mask = (index >> bit) & 1u;
        X ^= mask * kSobolDirections[dimension][bit];
    }
    return X;
}
// :193-205
static inline uint bhos_reverse_bits(uint x) {
    x = (((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1));
    x = (((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2));
    x = (((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4));
    x = (((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8));
    return ((x >> 16) | (x << 16));
}
// :207-223 (the psychopath.io LK hash variant is the live branch)
static inline uint bhos_owen_hash(uint x, uint seed) {
    x ^= x * 0x3d20adeau;
    x += seed;
    x *= (seed >> 16) | 1u;
    x ^= x * 0x05526c56u;
    x ^= x * 0x53a22864u;
    return x;
}
// :225-231
static inline uint bhos_owen_scramble(uint x, uint seed) {
    x = bhos_reverse_bits(x);
    x = bhos_owen_hash(x, seed);
    x = bhos_reverse_bits(x);
    return x;
}

// SampleGenerators.hlsli:16-24
enum SampleGeneratorEffectSeed : uint {
    SGES_Base = 0, SGES_ScatterBSDF = 1, SGES_NextEventEstimation = 2, SGES_NextEventEstimationLightSampler = 3,
    SGES_NextEventEstimationFeedback = 5, SGES_RussianRoulette = 6,
};
// SampleGenerators.hlsli:27
static const uint kDisableLowDiscrepancySamplingAfterDiffuseBounceCount = 1;

// StatelessSampleGenerators.hlsli:18-50
struct SampleGeneratorVertexBase {
    uint m_baseHash;
    uint m_sampleIndex;
    static SampleGeneratorVertexBase make(uint packedPixel, uint vertexIndex, uint sampleIndex) {
        SampleGeneratorVertexBase r;
        r.m_sampleIndex = sampleIndex;
        r.m_baseHash = Hash32Combine(Hash32(vertexIndex + 0x035F9F29u), packedPixel);
        return r;
    }
};

// StatelessSampleGenerators.hlsli:62-171 — the low-discrepancy / uniform dual-mode generator ("SampleGenerator")
struct SampleSequenceGenerator {
    static const uint cLDDisabled = 0xFFFFFFFEu;
    static const uint cLDDisabled_RanOutOfDimensions = 0xFFFFFFFFu;
    uint m_startingHash, m_currentHash, m_sampleIndex, m_dimension, m_activeIndex;

    static SampleSequenceGenerator make(const SampleGeneratorVertexBase& base, uint effectSeed = SGES_Base,
                                        bool lowDiscrepancy = false, int subSampleCount = 1) {
        SampleSequenceGenerator r;
        r.m_sampleIndex = base.m_sampleIndex;
        r.m_activeIndex = r.m_sampleIndex * (uint)subSampleCount;
        r.m_currentHash = Hash32Combine(base.m_baseHash, effectSeed);
        r.m_startingHash = r.m_currentHash;
        if (lowDiscrepancy) r.m_dimension = 0;
        else { r.m_currentHash = Hash32Combine(r.m_currentHash, r.m_activeIndex); r.m_dimension = cLDDisabled; }
        return r;
    }
    uint Next() {
        const uint maxSupportedDimensionIndex = 5;
        if (m_dimension >= cLDDisabled) { m_currentHash = Hash32(m_currentHash); return m_currentHash; }
        uint shuffle_seed = Hash32Combine(m_currentHash, 0);
        uint dim_seed = Hash32Combine(m_currentHash, 1 + m_dimension);
        uint shuffled_index = bhos_owen_scramble(m_activeIndex, shuffle_seed);
        uint dim_sample = (m_dimension == 0) ? bhos_reverse_bits(shuffled_index) : bhos_sobol(shuffled_index, m_dimension);
        dim_sample = bhos_owen_scramble(dim_sample, dim_seed);
        m_dimension++;
        if (m_dimension >= maxSupportedDimensionIndex) {
            m_currentHash = Hash32Combine(m_currentHash, m_activeIndex);
            m_dimension = cLDDisabled_RanOutOfDimensions;
        }
        return dim_sample;
    }
    // :146-171 — static Generate(count<=4, ...) always in LD mode
    static float4 Generate(uint count, const SampleGeneratorVertexBase& base, uint effectSeed,
                           int subSampleIndex = 0, int subSampleCount = 1) {
        if (count > 4) count = 4;
        float v[4] = {0, 0, 0, 0};
        uint activeIndex = base.m_sampleIndex * (uint)subSampleCount + (uint)subSampleIndex;
        uint currentHash = Hash32Combine(base.m_baseHash, effectSeed);
        for (uint dim = 0; dim < count; dim++) {
            uint shuffle_seed = Hash32Combine(currentHash, 0);
            uint dim_seed = Hash32Combine(currentHash, 1 + dim);
            uint shuffled_index = bhos_owen_scramble(activeIndex, shuffle_seed);
            uint dim_sample = (dim == 0) ? bhos_reverse_bits(shuffled_index) : bhos_sobol(shuffled_index, dim);
            dim_sample = bhos_owen_scramble(dim_sample, dim_seed);
            v[dim] = Hash32ToFloat(dim_sample);
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};

// StatelessSampleGenerators.hlsli:179-232
struct UniformSampleSequenceGenerator {
    uint m_currentHash;
    static UniformSampleSequenceGenerator make(const SampleGeneratorVertexBase& base, uint effectSeed = SGES_Base,
                                               int subSampleCount = 1) {
        UniformSampleSequenceGenerator r;
        uint activeIndex = base.m_sampleIndex * (uint)subSampleCount;
        r.m_currentHash = Hash32Combine(base.m_baseHash, effectSeed);
        r.m_currentHash = Hash32Combine(r.m_currentHash, activeIndex);
        return r;
    }
    uint Next() { m_currentHash = Hash32(m_currentHash); return m_currentHash; }
    static float4 Generate(uint count, const SampleGeneratorVertexBase& base, uint effectSeed,
                           int subSampleIndex = 0, int subSampleCount = 1) {
        if (count > 4) count = 4;
        float v[4] = {0, 0, 0, 0};
        uint activeIndex = base.m_sampleIndex * (uint)subSampleCount + (uint)subSampleIndex;
        uint h = Hash32Combine(base.m_baseHash, effectSeed);
        h = Hash32Combine(h, activeIndex);
        for (uint i = 0; i < count; i++) { h = Hash32(h); v[i] = Hash32ToFloat(h); }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
};

// SampleGenerators.hlsli:45-52
template <typename G> static inline float sampleNext1D(G& g) { uint bits = g.Next(); return (float)(bits >> 8) / 16777216.0f; }
template <typename G> static inline float2 sampleNext2D(G& g) { float2 s; s.x = sampleNext1D(g); s.y = sampleNext1D(g); return s; }

#pragma clang force_cuda_host_device end
} // namespace ptk
