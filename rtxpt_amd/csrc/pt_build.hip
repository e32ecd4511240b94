// mi355pt — GPU LBVH build / refit kernels. See pt_build.h for the pipeline.
#include "pt_build.h"
#include "pt_build_sah.h"
#include "pt_build_wide.h"
#include "pt_build_reinsert.h"
#include <chrono>
#include <cstdlib>
#include <vector>
#include <rocprim/rocprim.hpp>

#ifndef PT_PLOC_RADIUS
#define PT_PLOC_RADIUS 32      // PLOC search window to either side (tools/bvh_lab: SAH cost 154 / 147 / 143 for 8 / 16 / 32 on C3; Karras 283; on the GPU 32 is +1.7 % Mrays/s over 16, profiles/r02k_isa_experiments.txt)
#endif
#ifndef PT_RANGE_TABLE_MAX_TRIS
#define PT_RANGE_TABLE_MAX_TRIS (16u << 20)     // above this the n log n sparse table (32 B x n x log2 n) gives way to the ticket-based k_bounds
#endif

namespace ptk {

#define PT_HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

__device__ __forceinline__ uint enc_float(float f) { uint b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float dec_float(uint u) { uint b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; return __uint_as_float(b); }

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }

// scene bounds: the waves of a block meet in LDS, six atomics per BLOCK (one per wave and word were 264 k same-address atomics per word at 2.8 M triangles — they serialise
// in the L2 at ~10^8 per second and address: 3 ms of a 5 ms refit, the pattern k_shade had in round 3)
__device__ __forceinline__ void block_bounds(float3 mn, float3 mx, uint* __restrict__ sceneBounds) {
    __shared__ float red[16][6];
    const uint wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, waves = (blockDim.x + 63u) >> 6;
    float a = wave_min(mn.x), b = wave_min(mn.y), c = wave_min(mn.z), d = wave_max(mx.x), e = wave_max(mx.y), f = wave_max(mx.z);
    if (lane == 0u) { red[wave][0] = a; red[wave][1] = b; red[wave][2] = c; red[wave][3] = d; red[wave][4] = e; red[wave][5] = f; }
    __syncthreads();
    if (threadIdx.x < 6u) {
        float v = red[0][threadIdx.x];
        for (uint w = 1; w < waves; w++) v = threadIdx.x < 3u ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3u) atomicMin(&sceneBounds[threadIdx.x], enc_float(v)); else atomicMax(&sceneBounds[threadIdx.x], enc_float(v));
    }
}
__global__ void __launch_bounds__(256) k_tri_setup(DeviceScene sc, uint numTris, TriRecord* __restrict__ triWorld, uint* __restrict__ sceneBounds) {
    uint p = blockIdx.x * 256u + threadIdx.x;
    float3 mn = make_float3(3.0e38f), mx = make_float3(-3.0e38f);
    if (p < numTris) {
        uint2 pi = sc.primInfo[p];
        uint2 ig = sc.subInstToInstGeom[pi.x];
        const InstanceDesc& inst = sc.instances[ig.x];
        const GeometryDesc& g = sc.geometries[ig.y];
        const SubInstanceData& si = sc.subInstances[pi.x];
        const uint* idx = sc.indices + g.indexOffset + 3u * pi.y;
        const float* P = sc.positions;
        uint i0 = g.vertexOffset + idx[0], i1 = g.vertexOffset + idx[1], i2 = g.vertexOffset + idx[2];
        float3 p0 = xform_point(inst.transform, make_float3(P[3 * i0], P[3 * i0 + 1], P[3 * i0 + 2]));
        float3 p1 = xform_point(inst.transform, make_float3(P[3 * i1], P[3 * i1 + 1], P[3 * i1 + 2]));
        float3 p2 = xform_point(inst.transform, make_float3(P[3 * i2], P[3 * i2 + 1], P[3 * i2 + 2]));
        bool alphaTested = (si.FlagsAndAlphaInfo & SubInstanceData::Flags_AlphaTested) != 0;
        bool excl = (si.FlagsAndAlphaInfo & SubInstanceData::Flags_ExcludeFromNEE) != 0;
        triWorld[p] = tri_record(p0, p1, p2, p, (alphaTested ? 1u : 0u) | (excl ? 3u : 0u), 0.f);      // (the pad is set once the scene bounds are known: k_leaf_boxes / k_bounds)
        mn = min3v(p0, min3v(p1, p2)); mx = max3v(p0, max3v(p1, p2));
    }
    block_bounds(mn, mx, sceneBounds);
}
// ---- refit (pt_build.h): the flat source records, made once per build in leaf order ...
__global__ void __launch_bounds__(256) k_tri_src(DeviceScene sc, const TriRecord* __restrict__ triSorted, uint n, TriSrc* __restrict__ out) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint p = triSorted[i].prim;
    const uint2 pi = sc.primInfo[p], ig = sc.subInstToInstGeom[pi.x];
    const GeometryDesc& g = sc.geometries[ig.y];
    const uint* idx = sc.indices + g.indexOffset + 3u * pi.y;
    TriSrc r; r.instance = ig.x; r.i0 = g.vertexOffset + idx[0]; r.i1 = g.vertexOffset + idx[1]; r.i2 = g.vertexOffset + idx[2]; r.prim = p; r.flags = triSorted[i].flags; r._pad0 = r._pad1 = 0u;
    out[i] = r;
}
// ... and the world-space triangles of a refit straight from them, in leaf order (the arithmetic of k_tri_setup; the pad follows in k_refit8_level, once the scene bounds are known)
__global__ void __launch_bounds__(1024) k_refit_world(DeviceScene sc, const TriSrc* __restrict__ src, uint n, TriRecord* __restrict__ triSorted, uint* __restrict__ sceneBounds) {
    uint i = blockIdx.x * 1024u + threadIdx.x;
    float3 mn = make_float3(3.0e38f), mx = make_float3(-3.0e38f);
    if (i < n) {
        const uint4 a = reinterpret_cast<const uint4*>(src)[2 * (size_t)i], b = reinterpret_cast<const uint4*>(src)[2 * (size_t)i + 1];      // instance i0 i1 i2 | prim flags
        const InstanceDesc& inst = sc.instances[a.x];
        const float* P = sc.positions;
        float3 p0 = xform_point(inst.transform, make_float3(P[3 * a.y], P[3 * a.y + 1], P[3 * a.y + 2]));
        float3 p1 = xform_point(inst.transform, make_float3(P[3 * a.z], P[3 * a.z + 1], P[3 * a.z + 2]));
        float3 p2 = xform_point(inst.transform, make_float3(P[3 * a.w], P[3 * a.w + 1], P[3 * a.w + 2]));
        triSorted[i] = tri_record(p0, p1, p2, b.x, b.y, 0.f);
        mn = min3v(p0, min3v(p1, p2)); mx = max3v(p0, max3v(p1, p2));
    }
    block_bounds(mn, mx, sceneBounds);
}

__device__ __forceinline__ unsigned long long expand21(uint v) {
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ void __launch_bounds__(256) k_morton(const TriRecord* __restrict__ triWorld, uint numTris, const uint* __restrict__ sceneBounds,
                                                unsigned long long* __restrict__ keys, uint* __restrict__ prims) {
    uint p = blockIdx.x * 256u + threadIdx.x;
    if (p >= numTris) return;
    float3 mn = make_float3(dec_float(sceneBounds[0]), dec_float(sceneBounds[1]), dec_float(sceneBounds[2]));
    float3 mx = make_float3(dec_float(sceneBounds[3]), dec_float(sceneBounds[4]), dec_float(sceneBounds[5]));
    float3 ext = mx - mn;
    TriRecord tr = triWorld[p];
    const float3 v0 = tri_v0(tr), c = v0 + ((tri_v1(tr) - v0) + (tri_v2(tr) - v0)) * (1.0f / 3.0f);
    float sx = ext.x > 0.f ? (c.x - mn.x) / ext.x : 0.f, sy = ext.y > 0.f ? (c.y - mn.y) / ext.y : 0.f, sz = ext.z > 0.f ? (c.z - mn.z) / ext.z : 0.f;
    uint qx = (uint)fminf(fmaxf(sx * 2097152.0f, 0.0f), 2097151.0f), qy = (uint)fminf(fmaxf(sy * 2097152.0f, 0.0f), 2097151.0f), qz = (uint)fminf(fmaxf(sz * 2097152.0f, 0.0f), 2097151.0f);
    keys[p] = (expand21(qx) << 2) | (expand21(qy) << 1) | expand21(qz);
    prims[p] = p;
}

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long ki = keys[i], kj = keys[j];
    if (ki == kj) return 64 + __clz((uint)i ^ (uint)j);
    return __clzll((long long)(ki ^ kj));
}
// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012
__global__ void __launch_bounds__(256) k_karras(const unsigned long long* __restrict__ keys, int n, uint* __restrict__ childL, uint* __restrict__ childR,
                                                uint* __restrict__ parent, uint* __restrict__ leafParent, uint* __restrict__ rangeFirst, uint* __restrict__ rangeLast) {
    int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int deltaMin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > deltaMin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) if (delta(keys, n, i, i + (l + t) * d) > deltaMin) l += t;
    int j = i + l * d;
    int deltaNode = delta(keys, n, i, j);
    int s = 0;
    int div = 2; int t = (l + div - 1) / div;
    while (true) {
        if (delta(keys, n, i, i + (s + t) * d) > deltaNode) s += t;
        if (t == 1) break;
        div *= 2; t = (l + div - 1) / div;
    }
    int gamma = i + s * d + (d < 0 ? d : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    uint left = (lo == gamma) ? ((uint)gamma | BVH_LEAF_BIT) : (uint)gamma;
    uint right = (hi == gamma + 1) ? ((uint)(gamma + 1) | BVH_LEAF_BIT) : (uint)(gamma + 1);
    childL[i] = left; childR[i] = right; rangeFirst[i] = (uint)lo; rangeLast[i] = (uint)hi;
    if (left & BVH_LEAF_BIT) leafParent[gamma] = (uint)i; else parent[gamma] = (uint)i;
    if (right & BVH_LEAF_BIT) leafParent[gamma + 1] = (uint)i; else parent[gamma + 1] = (uint)i;
    if (i == 0) parent[0] = 0xFFFFFFFFu;
}

__device__ __forceinline__ float scene_pad_of(const uint* __restrict__ sceneBounds) {
    return scene_pad(make_float3(dec_float(sceneBounds[0]), dec_float(sceneBounds[1]), dec_float(sceneBounds[2])), make_float3(dec_float(sceneBounds[3]), dec_float(sceneBounds[4]), dec_float(sceneBounds[5])));
}
__global__ void __launch_bounds__(256) k_bounds(const TriRecord* __restrict__ triWorld, const uint* __restrict__ primsSorted, uint n, TriRecord* __restrict__ triSorted,
                                                const uint* __restrict__ childL, const uint* __restrict__ parent, const uint* __restrict__ leafParent, uint* __restrict__ tickets,
                                                float4* boxLmin, float4* boxLmax, float4* boxRmin, float4* boxRmax, const uint* __restrict__ sceneBounds) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    TriRecord tr = triWorld[primsSorted[i]];
    float3 mn, mx; tri_bounds(tr, mn, mx);
    tr.pad = tri_pad(mn, mx, scene_pad_of(sceneBounds));
    triSorted[i] = tr;
    if (n == 1) { boxLmin[0] = make_float4(mn.x, mn.y, mn.z, 0.f); boxLmax[0] = make_float4(mx.x, mx.y, mx.z, 0.f); return; }
    uint childRef = i | BVH_LEAF_BIT;
    uint node = leafParent[i];
    while (true) {
        bool isLeft = (childL[node] == childRef);
        if (isLeft) { boxLmin[node] = make_float4(mn.x, mn.y, mn.z, 0.f); boxLmax[node] = make_float4(mx.x, mx.y, mx.z, 0.f); }
        else        { boxRmin[node] = make_float4(mn.x, mn.y, mn.z, 0.f); boxRmax[node] = make_float4(mx.x, mx.y, mx.z, 0.f); }
        __threadfence();                                   // release my box before taking the ticket
        uint old = atomicAdd(&tickets[node], 1u);
        if (old == 0u) return;                             // first arrival: the sibling finishes this node
        __threadfence();                                   // acquire the sibling's box
        float4 omn = isLeft ? boxRmin[node] : boxLmin[node];
        float4 omx = isLeft ? boxRmax[node] : boxLmax[node];
        mn = min3v(mn, make_float3(omn.x, omn.y, omn.z)); mx = max3v(mx, make_float3(omx.x, omx.y, omx.z));
        if (node == 0u) return;
        childRef = node;
        node = parent[node];
    }
}


// ---- PLOC (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", TVCG 2018), the default builder:
// the Morton-sorted triangles are the initial clusters; every pass each cluster looks PT_PLOC_RADIUS neighbours to either side for the partner that
// gives the smallest merged surface area, mutual nearest neighbours merge, and the cluster list is compacted with a prefix sum (order preserved), until
// one cluster is left. Everything is deterministic: ties go to the lower index, node numbers come from the prefix sum, no atomics.
// Against the Karras tree over the same codes this halves the SAH cost on C3 (283 -> 143, tools/bvh_lab) because neighbours are chosen by the
// boxes they actually produce, not by the bit pattern of their centroids.
// Node ids while building: leaves 0..n-1 (Morton position), inner nodes n.. in creation order (children before parents; the last one is the root).
__global__ void __launch_bounds__(256) k_ploc_init(const TriRecord* __restrict__ triWorld, const uint* __restrict__ primsSorted, uint n, float4* __restrict__ cbMin, float4* __restrict__ cbMax,
                                                   uint* __restrict__ cl, uint* __restrict__ nodeCnt) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    TriRecord tr = triWorld[primsSorted[i]];
    float3 mn, mx; tri_bounds(tr, mn, mx);
    cbMin[i] = make_float4(mn.x, mn.y, mn.z, 0.f); cbMax[i] = make_float4(mx.x, mx.y, mx.z, 0.f);
    cl[i] = i; nodeCnt[i] = 1u;
}
__device__ __forceinline__ float merged_area(float4 amn, float4 amx, float4 bmn, float4 bmx) {
    float ex = fmaxf(amx.x, bmx.x) - fminf(amn.x, bmn.x), ey = fmaxf(amx.y, bmx.y) - fminf(amn.y, bmn.y), ez = fmaxf(amx.z, bmx.z) - fminf(amn.z, bmn.z);
    return ex * ey + ey * ez + ez * ex;                      // symmetric in (a, b) bit for bit: mutual nearest neighbours see the same number
}
// nearest neighbour inside the window; the block's clusters and a halo of PT_PLOC_RADIUS on either side are staged through LDS
__global__ void __launch_bounds__(256) k_ploc_nn(const float4* __restrict__ cbMin, const float4* __restrict__ cbMax, uint m, uint* __restrict__ nn) {
    __shared__ float4 smn[256 + 2 * PT_PLOC_RADIUS], smx[256 + 2 * PT_PLOC_RADIUS];
    const int base = (int)(blockIdx.x * 256u) - PT_PLOC_RADIUS;
    for (uint k = threadIdx.x; k < 256u + 2u * PT_PLOC_RADIUS; k += 256u) {
        int g = base + (int)k;
        if (g >= 0 && g < (int)m) { smn[k] = cbMin[g]; smx[k] = cbMax[g]; }
    }
    __syncthreads();
    const uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint li = threadIdx.x + PT_PLOC_RADIUS;
    const float4 amn = smn[li], amx = smx[li];
    const int lo = ((int)i - PT_PLOC_RADIUS < 0) ? 0 : (int)i - PT_PLOC_RADIUS, hi = ((int)i + PT_PLOC_RADIUS > (int)m - 1) ? (int)m - 1 : (int)i + PT_PLOC_RADIUS;
    float best = 3.0e38f; uint bj = i;
    for (int j = lo; j <= hi; j++) {
        if (j == (int)i) continue;
        const float a = merged_area(amn, amx, smn[j - base], smx[j - base]);
        if (a < best) { best = a; bj = (uint)j; }           // strict: ties stay with the lower index
    }
    nn[i] = bj;
}
// flags[i] = keep (low word: this slot survives into the next list) | creates (high word: this slot becomes a new inner node)
__global__ void __launch_bounds__(256) k_ploc_mark(const uint* __restrict__ nn, uint m, unsigned long long* __restrict__ flags) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint j = nn[i];
    const bool mutual = (j != i) && (nn[j] == i);
    const bool keep = !(mutual && i > j), creates = mutual && i < j;
    flags[i] = (keep ? 1ull : 0ull) | (creates ? (1ull << 32) : 0ull);
}
__global__ void __launch_bounds__(256) k_ploc_emit(const uint* __restrict__ cl, const float4* __restrict__ cbMin, const float4* __restrict__ cbMax, const uint* __restrict__ nn,
                                                   const unsigned long long* __restrict__ flags, const unsigned long long* __restrict__ offs, uint m, uint n, uint nodeBase,
                                                   uint* __restrict__ clN, float4* __restrict__ cbMinN, float4* __restrict__ cbMaxN, uint* __restrict__ childA, uint* __restrict__ childB,
                                                   uint* __restrict__ nodeCnt, uint* __restrict__ nodeParent, uint* __restrict__ counts) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const unsigned long long f = flags[i], o = offs[i];
    if (i == m - 1u) { counts[0] = (uint)o + (uint)(f & 1ull); counts[1] = (uint)(o >> 32) + (uint)(f >> 32); }      // next list length, nodes created in this pass
    if (!(f & 1ull)) return;
    const uint pos = (uint)o;
    if (f >> 32) {
        const uint j = nn[i], a = cl[i], b = cl[j], id = nodeBase + (uint)(o >> 32);
        childA[id - n] = a; childB[id - n] = b; nodeCnt[id] = nodeCnt[a] + nodeCnt[b];
        nodeParent[a] = id; nodeParent[b] = id | 0x80000000u;                       // bit 31: "I am the right child"
        const float4 amn = cbMin[i], amx = cbMax[i], bmn = cbMin[j], bmx = cbMax[j];
        clN[pos] = id;
        cbMinN[pos] = make_float4(fminf(amn.x, bmn.x), fminf(amn.y, bmn.y), fminf(amn.z, bmn.z), 0.f);
        cbMaxN[pos] = make_float4(fmaxf(amx.x, bmx.x), fmaxf(amx.y, bmx.y), fmaxf(amx.z, bmx.z), 0.f);
    } else { clN[pos] = cl[i]; cbMinN[pos] = cbMin[i]; cbMaxN[pos] = cbMax[i]; }
}
// depth-first position of every node's first leaf: walking up, every time the node sits in a right sub-tree the left sibling's leaves come first
__global__ void __launch_bounds__(256) k_ploc_first(uint n, const uint* __restrict__ nodeParent, const uint* __restrict__ childA, const uint* __restrict__ nodeCnt, uint* __restrict__ first) {
    uint x = blockIdx.x * 256u + threadIdx.x;
    if (x >= 2u * n - 1u) return;
    uint pos = 0u, node = x;
    const uint root = 2u * n - 2u;
    while (node != root) {
        const uint p = nodeParent[node], pid = p & 0x7FFFFFFFu;
        if (p >> 31) pos += nodeCnt[childA[pid - n]];
        node = pid;
    }
    first[x] = pos;
}
// final numbering, the one every later stage uses (as after k_karras): inner node 0 is the root, leaves are referenced by their depth-first position
__global__ void __launch_bounds__(256) k_ploc_finish(uint n, const uint* __restrict__ first, const uint* __restrict__ nodeParent, const uint* __restrict__ childA, const uint* __restrict__ childB,
                                                     const uint* __restrict__ nodeCnt, const uint* __restrict__ primsMorton, uint* __restrict__ primsFinal,
                                                     uint* __restrict__ childL, uint* __restrict__ childR, uint* __restrict__ parent, uint* __restrict__ leafParent,
                                                     uint* __restrict__ rangeFirst, uint* __restrict__ rangeLast) {
    uint x = blockIdx.x * 256u + threadIdx.x;
    if (x >= 2u * n - 1u) return;
    const uint root = 2u * n - 2u;
    if (x < n) { const uint pos = first[x]; primsFinal[pos] = primsMorton[x]; leafParent[pos] = root - (nodeParent[x] & 0x7FFFFFFFu); return; }
    const uint id = root - x, a = childA[x - n], b = childB[x - n], f = first[x];
    childL[id] = (a < n) ? (first[a] | BVH_LEAF_BIT) : (root - a);
    childR[id] = (b < n) ? (first[b] | BVH_LEAF_BIT) : (root - b);
    rangeFirst[id] = f; rangeLast[id] = f + nodeCnt[x] - 1u;
    parent[id] = (x == root) ? 0xFFFFFFFFu : (root - (nodeParent[x] & 0x7FFFFFFFu));
}

// ---- node bounds by range queries (see pt_build.h)
__global__ void __launch_bounds__(256) k_leaf_boxes(const TriRecord* __restrict__ triWorld, const uint* __restrict__ primsSorted, uint n, TriRecord* __restrict__ triSorted,
                                                    float4* __restrict__ rmin, float4* __restrict__ rmax, const uint* __restrict__ sceneBounds) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    TriRecord tr = triWorld[primsSorted[i]];
    float3 mn, mx; tri_bounds(tr, mn, mx);
    tr.pad = tri_pad(mn, mx, scene_pad_of(sceneBounds));      // the triangle's own padded box is the innermost box of the hit definition (pt_scene.h tri_box_accepts)
    triSorted[i] = tr;
    rmin[i] = make_float4(mn.x, mn.y, mn.z, 0.f); rmax[i] = make_float4(mx.x, mx.y, mx.z, 0.f);
}
__global__ void __launch_bounds__(256) k_range_level(uint n, uint half, const float4* __restrict__ pmin, const float4* __restrict__ pmax, float4* __restrict__ omin, float4* __restrict__ omax) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i + 2u * half > n) return;                          // entry i of this level covers [i, i + 2*half)
    float4 a = pmin[i], b = pmin[i + half], c = pmax[i], d = pmax[i + half];
    omin[i] = make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), 0.f);
    omax[i] = make_float4(fmaxf(c.x, d.x), fmaxf(c.y, d.y), fmaxf(c.z, d.z), 0.f);
}
__device__ __forceinline__ void range_box(const float4* __restrict__ rmin, const float4* __restrict__ rmax, uint n, uint a, uint b, float4& mn, float4& mx) {
    uint len = b - a + 1u, k = 31u - (uint)__clz((int)len);
    size_t base = (size_t)k * n; uint j = b + 1u - (1u << k);
    float4 m0 = rmin[base + a], m1 = rmin[base + j], x0 = rmax[base + a], x1 = rmax[base + j];
    mn = make_float4(fminf(m0.x, m1.x), fminf(m0.y, m1.y), fminf(m0.z, m1.z), 0.f);
    mx = make_float4(fmaxf(x0.x, x1.x), fmaxf(x0.y, x1.y), fmaxf(x0.z, x1.z), 0.f);
}
__global__ void __launch_bounds__(256) k_node_boxes(uint n, const uint* __restrict__ childL, const uint* __restrict__ rangeFirst, const uint* __restrict__ rangeLast,
                                                    const float4* __restrict__ rmin, const float4* __restrict__ rmax,
                                                    float4* __restrict__ boxLmin, float4* __restrict__ boxLmax, float4* __restrict__ boxRmin, float4* __restrict__ boxRmax) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (n == 1u) { if (i == 0u) { boxLmin[0] = rmin[0]; boxLmax[0] = rmax[0]; } return; }
    if (i >= n - 1u) return;
    const uint first = rangeFirst[i], last = rangeLast[i], cl = childL[i];
    const uint gamma = (cl & BVH_LEAF_BIT) ? (cl & 0x7FFFFFFFu) : rangeLast[cl];      // the left child covers [first, gamma] (leaves are in depth-first order for both builders)
    float4 mn, mx;
    range_box(rmin, rmax, n, first, gamma, mn, mx); boxLmin[i] = mn; boxLmax[i] = mx;
    range_box(rmin, rmax, n, gamma + 1u, last, mn, mx); boxRmin[i] = mn; boxRmax[i] = mx;
}

// node boxes are padded with the expression that pads the triangles' own boxes (tri_pad): the pad grows with the extent, so a node's padded box
// contains the padded boxes of everything below it — the containment the hit definition rests on
__device__ __forceinline__ void pad_box(float3& mn, float3& mx, float scenePad) {
    float pad = tri_pad(mn, mx, scenePad);
    mn = mn - make_float3(pad); mx = mx + make_float3(pad);
}
__global__ void __launch_bounds__(256) k_emit(uint n, const uint* __restrict__ childL, const uint* __restrict__ childR, const uint* __restrict__ rangeFirst,
                                              const uint* __restrict__ rangeLast, const float4* __restrict__ boxLmin, const float4* __restrict__ boxLmax,
                                              const float4* __restrict__ boxRmin, const float4* __restrict__ boxRmax, const uint* __restrict__ sceneBounds, BvhNode* __restrict__ nodes,
                                              const uint* __restrict__ absorb) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    float3 smn = make_float3(dec_float(sceneBounds[0]), dec_float(sceneBounds[1]), dec_float(sceneBounds[2]));
    float3 smx = make_float3(dec_float(sceneBounds[3]), dec_float(sceneBounds[4]), dec_float(sceneBounds[5]));
    float scenePad = scene_pad(smn, smx);
    if (n == 1) {
        if (i == 0) {
            BvhNode nd; float4 a = boxLmin[0], b = boxLmax[0];
            nd.lmin = make_float3(a.x, a.y, a.z); nd.lmax = make_float3(b.x, b.y, b.z); pad_box(nd.lmin, nd.lmax, scenePad);
            nd.rmin = nd.lmin; nd.rmax = nd.lmax; nd.left = BVH_LEAF_BIT | 0u; nd.right = BVH_EMPTY; nd._pad0 = nd._pad1 = 0;
            nodes[0] = nd;
        }
        return;
    }
    if (i >= n - 1) return;
    uint cnt = rangeLast[i] - rangeFirst[i] + 1u;
    if (i != 0u && cnt <= BVH_MAX_LEAF) return;            // collapsed into its parent's leaf reference
    uint refs[2] = {childL[i], childR[i]};
    uint outRef[2], open = 0u;                              // open bit k: the cost-driven collapse opens child k inside this node's wide node (BVH_BUILDER_SAH)
    for (int k = 0; k < 2; k++) {
        uint r = refs[k];
        if (r & BVH_LEAF_BIT) outRef[k] = BVH_LEAF_BIT | ((r & 0x7FFFFFFFu) << 3) | 0u;
        else {
            uint c = rangeLast[r] - rangeFirst[r] + 1u;
            outRef[k] = (c <= BVH_MAX_LEAF) ? (BVH_LEAF_BIT | (rangeFirst[r] << 3) | (c - 1u)) : r;
            if (absorb && c > BVH_MAX_LEAF && absorb[r]) open |= 1u << k;
        }
    }
    BvhNode nd; float4 a = boxLmin[i], b = boxLmax[i], c = boxRmin[i], d = boxRmax[i];
    nd.lmin = make_float3(a.x, a.y, a.z); nd.lmax = make_float3(b.x, b.y, b.z); nd.rmin = make_float3(c.x, c.y, c.z); nd.rmax = make_float3(d.x, d.y, d.z);
    pad_box(nd.lmin, nd.lmax, scenePad); pad_box(nd.rmin, nd.rmax, scenePad);
    nd.left = outRef[0]; nd.right = outRef[1]; nd._pad0 = open; nd._pad1 = 0;
    nodes[i] = nd;
}
__global__ void k_init_bounds(uint* sceneBounds) {
    if (threadIdx.x < 3) sceneBounds[threadIdx.x] = 0xFFFFFFFFu; else if (threadIdx.x < 6) sceneBounds[threadIdx.x] = 0u;
}

// ---- BVH2 -> BVH8 collapse (one thread per wide node of the current level; host loops over levels)
__device__ __forceinline__ float box_area(float3 mn, float3 mx) { float3 e = mx - mn; return e.x * e.y + e.y * e.z + e.z * e.x; }
__device__ __forceinline__ uint pow2_exp_ge(float s) {          // biased exponent eb with 2^(eb-127) >= s, eb in [1, 254]
    uint b = __float_as_uint(s);
    uint eb = (b >> 23) & 0xFFu;
    if (b & 0x7FFFFFu) eb++;
    if (eb < 1u) eb = 1u;
    if (eb > 254u) eb = 254u;
    return eb;
}
__device__ __forceinline__ float q_decode(float o, uint q, float s) { return fmaf((float)q, s, o); }   // identical expression in traversal (v_pk_fma_f32)
// header and quantised child bounds of a wide node from its n child boxes (the references are the caller's): origin = the union's minimum, per-axis power-of-two scales so
// that code 255 reaches the maximum, child codes rounded outwards and checked with the traversal's own decode expression
__device__ __forceinline__ void bvh8_scales(float3 mn, float3 mx, uint& ex, uint& ey, uint& ez) {
    ex = pow2_exp_ge((mx.x - mn.x) / 255.0f); ey = pow2_exp_ge((mx.y - mn.y) / 255.0f); ez = pow2_exp_ge((mx.z - mn.z) / 255.0f);
    // make sure code 255 reaches the node maximum under the decode arithmetic
    while (q_decode(mn.x, 255u, __uint_as_float(ex << 23)) < mx.x && ex < 254u) ex++;
    while (q_decode(mn.y, 255u, __uint_as_float(ey << 23)) < mx.y && ey < 254u) ey++;
    while (q_decode(mn.z, 255u, __uint_as_float(ez << 23)) < mx.z && ez < 254u) ez++;
}
__device__ __forceinline__ void bvh8_child_codes(float3 cmn, float3 cmx, float3 org, float sx, float sy, float sz, uint& q0, uint& q1) {
    float lo[3] = {cmn.x, cmn.y, cmn.z}, hi[3] = {cmx.x, cmx.y, cmx.z}, o[3] = {org.x, org.y, org.z}, sc3[3] = {sx, sy, sz};
    uint ql[3], qh[3];
    for (int a = 0; a < 3; a++) {
        float fl = floorf((lo[a] - o[a]) / sc3[a]); fl = fminf(fmaxf(fl, 0.f), 255.f); uint q = (uint)fl;
        while (q > 0u && q_decode(o[a], q, sc3[a]) > lo[a]) q--;
        ql[a] = q;
        float fh = ceilf((hi[a] - o[a]) / sc3[a]); fh = fminf(fmaxf(fh, 0.f), 255.f); q = (uint)fh;
        while (q < 255u && q_decode(o[a], q, sc3[a]) < hi[a]) q++;
        qh[a] = q;
    }
    q0 = ql[0] | (ql[1] << 8) | (ql[2] << 16) | (qh[0] << 24); q1 = qh[1] | (qh[2] << 8);
}
// child k goes to slot slotOf[k]; the other slots are empty
__device__ __forceinline__ void bvh8_pack(Bvh8Node& out, const float3* cmn, const float3* cmx, uint n, const uint* slotOf) {
    float3 mn = cmn[0], mx = cmx[0];
    for (uint k = 1; k < n; k++) { mn = min3v(mn, cmn[k]); mx = max3v(mx, cmx[k]); }
    uint ex, ey, ez; bvh8_scales(mn, mx, ex, ey, ez);
    float sx = __uint_as_float(ex << 23), sy = __uint_as_float(ey << 23), sz = __uint_as_float(ez << 23);
    out.ox = mn.x; out.oy = mn.y; out.oz = mn.z; out.exps = ex | (ey << 8) | (ez << 16) | (n << 24);
    out._pad[0] = __float_as_uint(sx); out._pad[1] = __float_as_uint(sy); out._pad[2] = __float_as_uint(sz); out._pad[3] = 0;      // the scales again, as floats (traversal reads these; exps stays for tools)
    for (uint k = 0; k < 8u; k++) { out.c[k].ref = BVH_EMPTY; out.c[k].q0 = 0x00FFFFFFu; out.c[k].q1 = 0u; }     // inverted box
    for (uint k = 0; k < n; k++) bvh8_child_codes(cmn[k], cmx[k], mn, sx, sy, sz, out.c[slotOf[k]].q0, out.c[slotOf[k]].q1);
}
__global__ void __launch_bounds__(128) k_collapse8(const BvhNode* __restrict__ nodes2, const uint* __restrict__ levelIn, uint nIn, uint* __restrict__ levelOut,
                                                   uint* __restrict__ counter, Bvh8Node* __restrict__ nodes8, uint costDriven) {
    uint i = blockIdx.x * 128u + threadIdx.x;
    if (i >= nIn) return;
    uint wide = levelIn[2 * i], r = levelIn[2 * i + 1];
    float3 cmn[8], cmx[8]; uint cref[8]; uint n = 0;
    uint openMask = 0u;                                     // costDriven: children the host's dynamic programme opens inside this wide node (BvhNode._pad0 of their parent)
    { BvhNode nd = nodes2[r];
      if (nd.left != BVH_EMPTY) { cmn[n] = nd.lmin; cmx[n] = nd.lmax; cref[n] = nd.left; if (nd._pad0 & 1u) openMask |= 1u << n; n++; }
      if (nd.right != BVH_EMPTY) { cmn[n] = nd.rmin; cmx[n] = nd.rmax; cref[n] = nd.right; if (nd._pad0 & 2u) openMask |= 1u << n; n++; } }
    while (n < 8u) {
        int best = -1;
        if (costDriven) { if (openMask) best = __ffs((int)openMask) - 1; }
        else { float bestA = -1.f; for (uint k = 0; k < n; k++) if (!(cref[k] & BVH_LEAF_BIT)) { float a = box_area(cmn[k], cmx[k]); if (a > bestA) { bestA = a; best = (int)k; } } }
        if (best < 0) break;
        BvhNode nd = nodes2[cref[best]];
        openMask &= ~(1u << best);
        cmn[best] = nd.lmin; cmx[best] = nd.lmax; cref[best] = nd.left; if (nd._pad0 & 1u) openMask |= 1u << best;
        cmn[n] = nd.rmin; cmx[n] = nd.rmax; cref[n] = nd.right; if (nd._pad0 & 2u) openMask |= 1u << n;
        n++;
    }
    uint nInner = 0;
    for (uint k = 0; k < n; k++) if (!(cref[k] & BVH_LEAF_BIT)) nInner++;
    uint wbase = 0, obase = 0;
    if (nInner) { wbase = atomicAdd(&counter[0], nInner); obase = atomicAdd(&counter[1], nInner); }
    uint slotOf[8]; for (uint k = 0; k < n; k++) slotOf[k] = k;      // children in collapse order (slots by octant — Ylitie et al. 2017 — were measured in round 4 and lost: +25 % on k_extend, profiles/r04u_octant_order_ab.txt)
    Bvh8Node out; bvh8_pack(out, cmn, cmx, n, slotOf);
    uint inner = 0;
    for (uint k = 0; k < n; k++) {
        if (cref[k] & BVH_LEAF_BIT) out.c[slotOf[k]].ref = cref[k];
        else { out.c[slotOf[k]].ref = wbase + inner; levelOut[2 * (obase + inner)] = wbase + inner; levelOut[2 * (obase + inner) + 1] = cref[k]; inner++; }
    }
    nodes8[wide] = out;
}
// refit of one level of the wide tree (pt_build.h): EIGHT lanes per wide node, one per child slot — the node is one coalesced 128-byte read, every lane forms its child's
// box (the triangles of a leaf child, the stored un-padded box of an inner child), the node's boxes meet over the eight lanes (xor-shuffles: min / max are exact in any
// order), every lane quantises and stores its own 12-byte slot. The child slots and references stay as the collapse left them.
__device__ __forceinline__ float grp8_min(float v) { v = fminf(v, __shfl_xor(v, 1)); v = fminf(v, __shfl_xor(v, 2)); return fminf(v, __shfl_xor(v, 4)); }
__device__ __forceinline__ float grp8_max(float v) { v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); return fmaxf(v, __shfl_xor(v, 4)); }
__global__ void __launch_bounds__(256) k_refit8_level(uint first, uint count, Bvh8Node* __restrict__ nodes8, TriRecord* __restrict__ tris, float4* __restrict__ boxMin, float4* __restrict__ boxMax,
                                                      const uint* __restrict__ sceneBounds) {
    const uint gi = blockIdx.x * 256u + threadIdx.x, i = gi >> 3, k = gi & 7u;
    if (i >= count) return;                                 // (whole groups of eight leave together)
    const uint w = first + i;
    const float scenePad = scene_pad_of(sceneBounds);
    uint* nd = reinterpret_cast<uint*>(nodes8 + w);
    const uint n = nd[3] >> 24, r = nd[4u + 3u * k];
    const bool valid = r != BVH_EMPTY;                      // (the children sit in the slots the collapse chose for them, not in the first n)
    float3 mn = make_float3(3.0e38f), mx = make_float3(-3.0e38f);
    if (valid) {
        if (r & BVH_LEAF_BIT) {
            const uint slot0 = (r & 0x7FFFFFFFu) >> 3, cnt = (r & 7u) + 1u;
            for (uint j = 0; j < cnt; j++) {
                const float4* tp = reinterpret_cast<const float4*>(tris + slot0 + j);
                const float4 ta = tp[0], tb = tp[1], tc = tp[2];      // x0 x1 x2 | y0 y1 y2 | z0 z1 z2 (pt_scene.h TriRecord)
                const float3 tmn = make_float3(fminf_(ta.x, fminf_(ta.y, ta.z)), fminf_(tb.x, fminf_(tb.y, tb.z)), fminf_(tc.x, fminf_(tc.y, tc.z)));
                const float3 tmx = make_float3(fmaxf_(ta.x, fmaxf_(ta.y, ta.z)), fmaxf_(tb.x, fmaxf_(tb.y, tb.z)), fmaxf_(tc.x, fmaxf_(tc.y, tc.z)));
                tris[slot0 + j].pad = tri_pad(tmn, tmx, scenePad);      // the triangle's own padded box: innermost box of the hit definition (k_leaf_boxes)
                mn = min3v(mn, tmn); mx = max3v(mx, tmx);
            }
        } else { const float4 a = boxMin[r], b = boxMax[r]; mn = make_float3(a.x, a.y, a.z); mx = make_float3(b.x, b.y, b.z); }
    }
    const float3 umn = make_float3(grp8_min(mn.x), grp8_min(mn.y), grp8_min(mn.z)), umx = make_float3(grp8_max(mx.x), grp8_max(mx.y), grp8_max(mx.z));
    if (k == 0u) { boxMin[w] = make_float4(umn.x, umn.y, umn.z, 0.f); boxMax[w] = make_float4(umx.x, umx.y, umx.z, 0.f); }
    if (valid) pad_box(mn, mx, scenePad);
    const float3 nmn = make_float3(grp8_min(mn.x), grp8_min(mn.y), grp8_min(mn.z)), nmx = make_float3(grp8_max(mx.x), grp8_max(mx.y), grp8_max(mx.z));
    uint ex, ey, ez; bvh8_scales(nmn, nmx, ex, ey, ez);
    const float sx = __uint_as_float(ex << 23), sy = __uint_as_float(ey << 23), sz = __uint_as_float(ez << 23);
    uint q0 = 0x00FFFFFFu, q1 = 0u;                         // an empty slot: inverted box
    if (valid) bvh8_child_codes(mn, mx, nmn, sx, sy, sz, q0, q1);
    nd[4u + 3u * k] = valid ? r : BVH_EMPTY; nd[5u + 3u * k] = q0; nd[6u + 3u * k] = q1;
    if (k == 0u) { nd[0] = __float_as_uint(nmn.x); nd[1] = __float_as_uint(nmn.y); nd[2] = __float_as_uint(nmn.z); nd[3] = ex | (ey << 8) | (ez << 16) | (n << 24); }
    if (k == 1u) { nd[28] = __float_as_uint(sx); nd[29] = __float_as_uint(sy); nd[30] = __float_as_uint(sz); nd[31] = 0u; }
}

// AlphaTestImpl's inputs (BridgeDonut:929-971) gathered once per triangle instead of once per candidate hit
__global__ void __launch_bounds__(256) k_alpha_records(DeviceScene sc, const TriRecord* __restrict__ triSorted, uint n, AlphaRec* __restrict__ recs, uint* __restrict__ primToSlot) {
    uint i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    primToSlot[triSorted[i].prim] = i;
    AlphaRec r; r.t0 = r.t1 = r.t2 = make_float2(0.f, 0.f); r.tex = 0xFFFFFFFFu; r.cutoff = 0.f; r.wh = r.plane = r.fmt = r._pad = 0u;
    TriRecord tr = triSorted[i];
    if (tr.flags & 1u) {
        uint2 pi = sc.primInfo[tr.prim];
        const SubInstanceData& si = sc.subInstances[pi.x];
        if (si.FlagsAndAlphaInfo & SubInstanceData::Flags_AlphaTested) {
            const uint* idx = sc.indices + si.IndexOffset + pi.y * 3;
            r.t0 = sc.uvs[si.TexCoord1Offset + idx[0]]; r.t1 = sc.uvs[si.TexCoord1Offset + idx[1]]; r.t2 = sc.uvs[si.TexCoord1Offset + idx[2]];
            r.tex = si.AlphaTextureIndex(); r.cutoff = si.AlphaCutoff();
            const AlphaPlane ap = sc.alphaPlanes[r.tex]; r.wh = ap.wh; r.plane = ap.offset; r.fmt = ap.fmt;
        }
    }
    recs[i] = r;
}

// the flat shading records (pt_scene.h ShadeTri): one thread per global primitive walks the chain loadSurface used to walk per hit
__global__ void __launch_bounds__(256) k_shade_tris(DeviceScene sc, uint firstPrim, uint numTris, ShadeTri* __restrict__ out) {
    uint p = blockIdx.x * 256u + threadIdx.x;
    if (p >= numTris) return;
    p += firstPrim;
    const uint2 pi = sc.primInfo[p];
    const uint2 ig = sc.subInstToInstGeom[pi.x];
    const GeometryDesc& g = sc.geometries[ig.y];
    const uint* idx = sc.indices + g.indexOffset + 3u * pi.y;
    const uint v0 = g.vertexOffset + idx[0], v1 = g.vertexOffset + idx[1], v2 = g.vertexOffset + idx[2];
    const float* P = sc.positions;
    ShadeTri r; __builtin_memset(&r, 0, sizeof(r));
    r.instance = ig.x; r.subInstance = pi.x; r.triangleIndex = pi.y;
    r.materialAndFlags = (sc.subInstances[pi.x].GlobalGeometryIndex_PTMaterialDataIndex & 0xFFFFu) | (g.flags << 16);
    r.p0 = make_float3(P[3 * v0], P[3 * v0 + 1], P[3 * v0 + 2]); r.p1 = make_float3(P[3 * v1], P[3 * v1 + 1], P[3 * v1 + 2]); r.p2 = make_float3(P[3 * v2], P[3 * v2 + 1], P[3 * v2 + 2]);
    if (g.flags & GEOM_HAS_UV) { r.t0 = sc.uvs[v0]; r.t1 = sc.uvs[v1]; r.t2 = sc.uvs[v2]; }
    if (g.flags & GEOM_HAS_NORMAL) {      // exactly loadSurface's expressions (pt_path.h, the PT_SHADE_TRI == 0 branch keeps them): the same floats, formed once
        const float3 objFlatN = SafeNormalize(cross(r.p1 - r.p0, r.p2 - r.p0));
        const uint pn[3] = {sc.normals[v0], sc.normals[v1], sc.normals[v2]}; float3 n[3];
        for (int k = 0; k < 3; k++) { n[k] = normalize(Unpack_RGB8_SNORM(pn[k])); if (dot(n[k], objFlatN) < 0.f) n[k] = -n[k]; }
        r.n0 = n[0]; r.n1 = n[1]; r.n2 = n[2];
    }
    if (g.flags & GEOM_HAS_TANGENT) { r.g0 = sc.tangents[v0]; r.g1 = sc.tangents[v1]; r.g2 = sc.tangents[v2]; }
    uint4* o = reinterpret_cast<uint4*>(out + p); const uint4* src = reinterpret_cast<const uint4*>(&r);
    for (int k = 0; k < 8; k++) o[k] = src[k];
}
void launch_shade_tris(const DeviceScene& sc, uint firstPrim, uint numTris, ShadeTri* out, hipStream_t st) {
    if (numTris) hipLaunchKernelGGL(k_shade_tris, dim3((numTris + 255u) / 256u), dim3(256), 0, st, sc, firstPrim, numTris, out);
}

// ---- cost-driven wide-node assignment on the device (pt_build_wide.h): the inner nodes above the wide tree's leaves are numbered breadth first (one launch per level,
// children appended with one atomic per block), the dynamic programme runs over the levels bottom-up, the marking top-down. ~3 launches per level of a tree ~50 deep.
__global__ void __launch_bounds__(256) k_wide_levels(const uint* __restrict__ levelIn, uint nIn, const uint* __restrict__ childL, const uint* __restrict__ childR,
                                                     const uint* __restrict__ rangeFirst, const uint* __restrict__ rangeLast, uint maxLeaf, uint* __restrict__ levelOut, uint* __restrict__ counter) {
    __shared__ uint sCnt[4]; __shared__ uint sBase;
    const uint i = blockIdx.x * 256u + threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint kids[2]; uint nk = 0;
    if (i < nIn) {
        const uint id = levelIn[i];
        if (rangeLast[id] - rangeFirst[id] + 1u > maxLeaf) {          // below a leaf of the wide tree nothing is decided
            const uint L = childL[id], R = childR[id];
            if (!(L & BVH_LEAF_BIT)) kids[nk++] = L;
            if (!(R & BVH_LEAF_BIT)) kids[nk++] = R;
        }
    }
    // exclusive prefix of nk over the block: per-wave by two ballots, per-block through LDS
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(nk >= 1u), m2 = __builtin_amdgcn_ballot_w64(nk >= 2u), below = (1ull << lane) - 1ull;
    const uint inWave = (uint)__popcll(m1 & below) + (uint)__popcll(m2 & below);
    if (lane == 0u) sCnt[wave] = (uint)__popcll(m1) + (uint)__popcll(m2);
    __syncthreads();
    if (threadIdx.x == 0u) { uint tot = 0; for (uint w = 0; w < 4u; w++) { const uint c = sCnt[w]; sCnt[w] = tot; tot += c; } sBase = tot ? atomicAdd(counter, tot) : 0u; }
    __syncthreads();
    const uint at = sBase + sCnt[wave] + inWave;
    for (uint k = 0; k < nk; k++) levelOut[at + k] = kids[k];
}
__global__ void __launch_bounds__(256) k_wide_dp(const uint* __restrict__ level, uint nIn, const uint* __restrict__ childL, const uint* __restrict__ childR, const uint* __restrict__ rangeFirst,
                                                 const uint* __restrict__ rangeLast, uint maxLeaf, const float4* __restrict__ boxLmin, const float4* __restrict__ boxLmax,
                                                 const float4* __restrict__ boxRmin, const float4* __restrict__ boxRmax, float* __restrict__ C, unsigned long long* __restrict__ dec) {
    const uint i = blockIdx.x * 256u + threadIdx.x; if (i >= nIn) return;
    const uint id = level[i];
    const float4 a = boxLmin[id], b = boxLmax[id], c = boxRmin[id], d = boxRmax[id];
    const float lmn[3] = {a.x, a.y, a.z}, lmx[3] = {b.x, b.y, b.z}, rmn[3] = {c.x, c.y, c.z}, rmx[3] = {d.x, d.y, d.z};
    wide_dp_node(id, childL[id], childR[id], rangeLast[id] - rangeFirst[id] + 1u, maxLeaf, lmn, lmx, rmn, rmx, C, dec);
}
__global__ void __launch_bounds__(256) k_wide_mark(const uint* __restrict__ level, uint nIn, const uint* __restrict__ childL, const uint* __restrict__ childR, const uint* __restrict__ rangeFirst,
                                                   const uint* __restrict__ rangeLast, uint maxLeaf, const unsigned long long* __restrict__ dec, uint* __restrict__ absorb, uint* __restrict__ state) {
    const uint i = blockIdx.x * 256u + threadIdx.x; if (i >= nIn) return;
    const uint id = level[i];
    wide_mark_node(id, state[id], childL[id], childR[id], rangeLast[id] - rangeFirst[id] + 1u, maxLeaf, dec, absorb, state);
}
// needs the child boxes of k_node_boxes (boxLmin .. boxRmax); scratch: levelA (breadth-first order), levelB (states), the dead sparse table (C), plocFlags (decisions)
static hipError_t bvh_wide_nodes(BvhBuildBuffers& b, uint n, hipStream_t st) {
    if (n < 2u || !b.rangeMin) return hipErrorInvalidValue;
    uint* order = b.levelA; uint* state = b.levelB; float* C = reinterpret_cast<float*>(b.rangeMin); unsigned long long* dec = b.plocFlags;
    PT_HIP_TRY(hipMemsetAsync(b.absorb, 0, 4 * (size_t)n, st)); PT_HIP_TRY(hipMemsetAsync(state, 0, 4 * (size_t)n, st));
    const uint first[2] = {0u, wide_state(8u, true)};                          // order[0] = the root; its state: a wide node with the whole budget
    PT_HIP_TRY(hipMemcpyAsync(order, &first[0], 4, hipMemcpyHostToDevice, st)); PT_HIP_TRY(hipMemcpyAsync(state, &first[1], 4, hipMemcpyHostToDevice, st));
    std::vector<uint> start{0u}, count{1u};
    for (;;) {                                                                  // breadth-first numbering, one level per launch (the level size comes back: a build step)
        const uint s0 = start.back(), c0 = count.back();
        if ((size_t)s0 + c0 > n) return hipErrorUnknown;
        PT_HIP_TRY(hipMemsetAsync(b.wideCounter, 0, 4, st));
        hipLaunchKernelGGL(k_wide_levels, dim3((c0 + 255u) / 256u), dim3(256), 0, st, order + s0, c0, b.childL, b.childR, b.rangeFirst, b.rangeLast, BVH_MAX_LEAF, order + s0 + c0, b.wideCounter);
        uint next = 0; PT_HIP_TRY(hipMemcpyAsync(&next, b.wideCounter, 4, hipMemcpyDeviceToHost, st)); PT_HIP_TRY(hipStreamSynchronize(st));
        if (!next) break;
        start.push_back(s0 + c0); count.push_back(next);
        if (start.size() > 100000u) return hipErrorUnknown;
    }
    for (size_t d = start.size(); d-- > 0;)
        hipLaunchKernelGGL(k_wide_dp, dim3((count[d] + 255u) / 256u), dim3(256), 0, st, order + start[d], count[d], b.childL, b.childR, b.rangeFirst, b.rangeLast, BVH_MAX_LEAF, b.boxLmin, b.boxLmax, b.boxRmin, b.boxRmax, C, dec);
    for (size_t d = 0; d < start.size(); d++)
        hipLaunchKernelGGL(k_wide_mark, dim3((count[d] + 255u) / 256u), dim3(256), 0, st, order + start[d], count[d], b.childL, b.childR, b.rangeFirst, b.rangeLast, BVH_MAX_LEAF, dec, b.absorb, state);
    b.wideLevels = (uint)start.size();
    return hipGetLastError();
}

static hipError_t bvh_alloc_all(BvhBuildBuffers& b, uint numTris);
hipError_t bvh_alloc(BvhBuildBuffers& b, uint numTris) {
    hipError_t e = bvh_alloc_all(b, numTris);
    if (e != hipSuccess) bvh_free(b);                        // no partial allocations survive a failure
    return e;
}
static hipError_t bvh_alloc_all(BvhBuildBuffers& b, uint numTris) {
    __builtin_memset(&b, 0, sizeof(b));
    uint n = numTris < 2 ? 2 : numTris;
    b.capacity = n;
    PT_HIP_TRY(hipMalloc(&b.triWorld, sizeof(TriRecord) * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.triSorted, sizeof(TriRecord) * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.keys, 8 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.keysSorted, 8 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.prims, 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.primsSorted, 4 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.childL, 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.childR, 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.parent, 4 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.leafParent, 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.rangeFirst, 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.rangeLast, 4 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.tickets, 4 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.boxLmin, 16 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.boxLmax, 16 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.boxRmin, 16 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.boxRmax, 16 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.sceneBounds, 32)); PT_HIP_TRY(hipMalloc(&b.nodes, sizeof(BvhNode) * (size_t)n));
    if (n <= PT_RANGE_TABLE_MAX_TRIS) { b.rangeLevels = 32u - (uint)__builtin_clz(n); PT_HIP_TRY(hipMalloc(&b.rangeMin, 16 * (size_t)n * b.rangeLevels)); PT_HIP_TRY(hipMalloc(&b.rangeMax, 16 * (size_t)n * b.rangeLevels)); }
    PT_HIP_TRY(hipMalloc(&b.triSrc, sizeof(TriSrc) * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.wideBoxMin, 16 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.wideBoxMax, 16 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.alphaRecs, sizeof(AlphaRec) * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.primToSlot, 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.nodes8, sizeof(Bvh8Node) * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.levelA, 8 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.levelB, 8 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.wideCounter, 16));
    size_t tmp = 0;
    PT_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, b.keys, b.keysSorted, b.prims, b.primsSorted, (size_t)n, 0, 64));
    b.sortTempBytes = tmp; PT_HIP_TRY(hipMalloc(&b.sortTemp, tmp ? tmp : 16));
    // PLOC work arrays (cluster boxes ping-pong through boxLmin/boxLmax and boxRmin/boxRmax, which the bounds stage overwrites afterwards)
    PT_HIP_TRY(hipMalloc(&b.plocCl[0], 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.plocCl[1], 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.plocNN, 4 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.plocFlags, 8 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.plocOffs, 8 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.plocChildA, 4 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.plocChildB, 4 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.plocCnt, 8 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.plocParent, 8 * (size_t)n)); PT_HIP_TRY(hipMalloc(&b.plocFirst, 8 * (size_t)n));
    PT_HIP_TRY(hipMalloc(&b.plocCounts, 16)); PT_HIP_TRY(hipMalloc(&b.absorb, 4 * (size_t)n));
    b.riScratchBytes = 176ull * (size_t)n + (64u << 10); PT_HIP_TRY(hipMalloc(&b.riScratch, b.riScratchBytes));      // parallel re-insertion (bvh_reinsert): ~160 bytes per triangle
    tmp = 0;
    PT_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, b.plocFlags, b.plocOffs, 0ull, (size_t)n, rocprim::plus<unsigned long long>()));
    b.scanTempBytes = tmp; PT_HIP_TRY(hipMalloc(&b.scanTemp, tmp ? tmp : 16));
    return hipSuccess;
}
void bvh_free(BvhBuildBuffers& b) {
    void* ps[] = {b.triWorld, b.triSorted, b.keys, b.keysSorted, b.prims, b.primsSorted, b.childL, b.childR, b.parent, b.leafParent, b.rangeFirst, b.rangeLast, b.tickets,
                  b.boxLmin, b.boxLmax, b.boxRmin, b.boxRmax, b.sceneBounds, b.nodes, b.sortTemp, b.nodes8, b.levelA, b.levelB, b.wideCounter, b.alphaRecs, b.primToSlot, b.rangeMin, b.rangeMax,
                  b.plocCl[0], b.plocCl[1], b.plocNN, b.plocFlags, b.plocOffs, b.plocChildA, b.plocChildB, b.plocCnt, b.plocParent, b.plocFirst, b.plocCounts, b.scanTemp, b.absorb, b.riScratch, b.triSrc, b.wideBoxMin, b.wideBoxMax};
    for (void* p : ps) if (p) (void)hipFree(p);
    __builtin_memset(&b, 0, sizeof(b));
}
// bottom-up over the levels of the wide tree: children lie in deeper collapse levels than their parents, and a level's nodes are one index range
static void bvh_refit_levels(BvhBuildBuffers& b, hipStream_t st) {
    for (uint l = b.collapseLevels; l-- > 0u;) {
        const uint first = b.wideLevelStart[l], count = b.wideLevelStart[l + 1u] - first;
        if (count) hipLaunchKernelGGL(k_refit8_level, dim3((count + 31u) / 32u), dim3(256), 0, st, first, count, b.nodes8, b.triSorted, b.wideBoxMin, b.wideBoxMax, b.sceneBounds);
    }
}
static hipError_t bvh_bounds_and_emit(BvhBuildBuffers& b, const DeviceScene& sc, uint n, hipStream_t st) {
    uint g = (n + 255u) / 256u;
    if (b.rangeMin) {
        hipLaunchKernelGGL(k_leaf_boxes, dim3(g), dim3(256), 0, st, b.triWorld, b.primsSorted, n, b.triSorted, b.rangeMin, b.rangeMax, b.sceneBounds);
        for (uint k = 1; k < b.rangeLevels && (1u << k) <= n; k++)
            hipLaunchKernelGGL(k_range_level, dim3(g), dim3(256), 0, st, n, 1u << (k - 1), b.rangeMin + (size_t)(k - 1) * n, b.rangeMax + (size_t)(k - 1) * n, b.rangeMin + (size_t)k * n, b.rangeMax + (size_t)k * n);
        hipLaunchKernelGGL(k_node_boxes, dim3(g), dim3(256), 0, st, n, b.childL, b.rangeFirst, b.rangeLast, b.rangeMin, b.rangeMax, b.boxLmin, b.boxLmax, b.boxRmin, b.boxRmax);
    } else {
        PT_HIP_TRY(hipMemsetAsync(b.tickets, 0, 4 * (size_t)(n < 2 ? 2 : n), st));
        hipLaunchKernelGGL(k_bounds, dim3(g), dim3(256), 0, st, b.triWorld, b.primsSorted, n, b.triSorted, b.childL, b.parent, b.leafParent, b.tickets, b.boxLmin, b.boxLmax, b.boxRmin, b.boxRmax, b.sceneBounds);
    }
    hipLaunchKernelGGL(k_alpha_records, dim3(g), dim3(256), 0, st, sc, b.triSorted, n, b.alphaRecs, b.primToSlot);
    if (b.wideDpPending) {                                  // BVH_BUILDER_PLOC_OPT build: the wide-node programme over the boxes just written; without the sparse table (huge scenes) the collapse stays greedy
        b.wideDpPending = 0u;
        if (n > 1u && b.rangeMin && bvh_wide_nodes(b, n, st) == hipSuccess) b.wideFlagsValid = 1u; else { (void)hipGetLastError(); b.wideFlagsValid = 0u; }
    }
    const bool costDriven = ((b.builder == BVH_BUILDER_SAH) || (b.builder == BVH_BUILDER_PLOC_OPT && b.wideFlagsValid)) && b.absorb != nullptr;      // (a refit finds the builder that made the topology)
    hipLaunchKernelGGL(k_emit, dim3(g), dim3(256), 0, st, n, b.childL, b.childR, b.rangeFirst, b.rangeLast, b.boxLmin, b.boxLmax, b.boxRmin, b.boxRmax, b.sceneBounds, b.nodes,
                       costDriven ? b.absorb : (const uint*)nullptr);
    // BVH8 collapse, level by level (the per-level node count comes back to the host: a build step, not the hot path)
    uint init[4] = {0u, 0u, 1u, 0u};                       // levelA[0] = (wide 0, bvh2 root 0); counter = {next wide index = 1, out count = 0}
    PT_HIP_TRY(hipMemcpyAsync(b.levelA, init, 8, hipMemcpyHostToDevice, st));
    PT_HIP_TRY(hipMemcpyAsync(b.wideCounter, init + 2, 8, hipMemcpyHostToDevice, st));
    uint nIn = 1, levels = 0; uint* in = b.levelA; uint* out = b.levelB;
    b.wideLevelStart[0] = 0u; b.wideLevelStart[1] = 1u;
    while (nIn) {
        hipLaunchKernelGGL(k_collapse8, dim3((nIn + 127u) / 128u), dim3(128), 0, st, b.nodes, in, nIn, out, b.wideCounter, b.nodes8, costDriven ? 1u : 0u);
        uint host[2];
        PT_HIP_TRY(hipMemcpyAsync(host, b.wideCounter, 8, hipMemcpyDeviceToHost, st));
        PT_HIP_TRY(hipStreamSynchronize(st));
        b.numNodes8 = host[0]; nIn = host[1];
        if (levels + 2u <= BVH_MAX_WIDE_LEVELS) b.wideLevelStart[levels + 2u] = host[0];      // the wide nodes the level just allocated: the next level, [wideLevelStart[levels + 1], host[0])
        uint zero = 0; PT_HIP_TRY(hipMemcpyAsync(b.wideCounter + 1, &zero, 4, hipMemcpyHostToDevice, st));
        uint* t = in; in = out; out = t;
        if (++levels > 4096u) return hipErrorUnknown;
    }
    b.collapseLevels = levels;
    // what a refit needs (pt_build.h): the flat source records in leaf order and every wide node's un-padded box — the latter by running the refit's own level pass once over the
    // tree just written (it rewrites the nodes with the bytes they hold: min / max are exact)
    b.wideRefitReady = 0u;
    if (levels <= BVH_MAX_WIDE_LEVELS && b.triSrc) {
        hipLaunchKernelGGL(k_tri_src, dim3(g), dim3(256), 0, st, sc, b.triSorted, n, b.triSrc);
        bvh_refit_levels(b, st);
        b.wideRefitReady = 1u;
    }
    return hipGetLastError();
}
static hipError_t bvh_reinsert(BvhBuildBuffers& b, uint n, uint passes, hipStream_t st);
// PLOC passes; every pass reads the new list length back (a build step, not the hot path: ~70 passes of 4 small launches at 2.8 M triangles)
static hipError_t bvh_ploc(BvhBuildBuffers& b, uint n, hipStream_t st) {
    float4* cbMin[2] = {b.boxLmin, b.boxRmin}; float4* cbMax[2] = {b.boxLmax, b.boxRmax};
    hipLaunchKernelGGL(k_ploc_init, dim3((n + 255u) / 256u), dim3(256), 0, st, b.triWorld, b.primsSorted, n, cbMin[0], cbMax[0], b.plocCl[0], b.plocCnt);
    uint m = n, nodeBase = n, cur = 0, passes = 0;
    while (m > 1u) {
        const uint g = (m + 255u) / 256u;
        hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(256), 0, st, cbMin[cur], cbMax[cur], m, b.plocNN);
        hipLaunchKernelGGL(k_ploc_mark, dim3(g), dim3(256), 0, st, b.plocNN, m, b.plocFlags);
        size_t tmp = b.scanTempBytes;
        PT_HIP_TRY(rocprim::exclusive_scan(b.scanTemp, tmp, b.plocFlags, b.plocOffs, 0ull, (size_t)m, rocprim::plus<unsigned long long>(), st));
        hipLaunchKernelGGL(k_ploc_emit, dim3(g), dim3(256), 0, st, b.plocCl[cur], cbMin[cur], cbMax[cur], b.plocNN, b.plocFlags, b.plocOffs, m, n, nodeBase,
                           b.plocCl[cur ^ 1u], cbMin[cur ^ 1u], cbMax[cur ^ 1u], b.plocChildA, b.plocChildB, b.plocCnt, b.plocParent, b.plocCounts);
        uint host[2] = {0u, 0u};
        PT_HIP_TRY(hipMemcpyAsync(host, b.plocCounts, 8, hipMemcpyDeviceToHost, st));
        PT_HIP_TRY(hipStreamSynchronize(st));
        if (host[0] >= m || host[1] == 0u) return hipErrorUnknown;            // every pass merges at least the globally closest pair
        m = host[0]; nodeBase += host[1]; cur ^= 1u;
        if (++passes > 100000u) return hipErrorUnknown;
    }
    b.plocPasses = passes;
    if (nodeBase != 2u * n - 1u) return hipErrorUnknown;
    if (b.builder == BVH_BUILDER_PLOC_OPT && b.riPasses && bvh_reinsert(b, n, b.riPasses, st) != hipSuccess) {      // insertion-based optimisation of the finished tree ("prefer fast trace")
        (void)hipGetLastError(); b.optimiserPasses = 0u;      // (a tree deeper than RI_MAX_LEVELS: the optimiser works on copies until its last kernel, the PLOC tree is intact and is used as it is)
    }
    const uint gAll = (2u * n - 1u + 255u) / 256u;
    hipLaunchKernelGGL(k_ploc_first, dim3(gAll), dim3(256), 0, st, n, b.plocParent, b.plocChildA, b.plocCnt, b.plocFirst);
    hipLaunchKernelGGL(k_ploc_finish, dim3(gAll), dim3(256), 0, st, n, b.plocFirst, b.plocParent, b.plocChildA, b.plocChildB, b.plocCnt, b.primsSorted, b.prims,
                       b.childL, b.childR, b.parent, b.leafParent, b.rangeFirst, b.rangeLast);
    PT_HIP_TRY(hipMemcpyAsync(b.primsSorted, b.prims, 4 * (size_t)n, hipMemcpyDeviceToDevice, st));      // leaf order = depth-first order from here on
    return hipGetLastError();
}
// ---- parallel re-insertion on the PLOC tree (pt_build_reinsert.h), in PLOC's node numbering: leaves 0 .. n - 1 in Morton order, inner nodes n .. 2n - 2, root 2n - 2.
// Scratch (b.riScratch, 160 bytes per triangle): parent / left / right, boxes, per-node proposals (gain, target, pivot), locks, the keys of the moving nodes, a
// breadth-first numbering of the tree (rebuilt every pass: it orders the refit) and its level bounds.
struct RiDevice { RiTree t; float* gain; uint* target; uint* pivot; unsigned long long* lock; unsigned long long* moving; uint* ok; uint* bfs; uint* bounds; uint* cursor; };
static const uint RI_MAX_LEVELS = 1024;
__global__ void __launch_bounds__(256) k_ri_init(uint n, const uint* __restrict__ nodeParent, const uint* __restrict__ childA, const uint* __restrict__ childB,
                                                 const TriRecord* __restrict__ triWorld, const uint* __restrict__ primsMorton, RiTree t) {
    const uint x = blockIdx.x * 256u + threadIdx.x; if (x >= t.N) return;
    t.par[x] = (x == t.N - 1u) ? RI_NONE : (nodeParent[x] & 0x7FFFFFFFu);
    if (x < n) {
        t.left[x] = RI_NONE; t.right[x] = RI_NONE;
        const TriRecord tr = triWorld[primsMorton[x]];
        float3 mn, mx; tri_bounds(tr, mn, mx);
        RiBox b; b.mn[0] = mn.x; b.mn[1] = mn.y; b.mn[2] = mn.z; b.mx[0] = mx.x; b.mx[1] = mx.y; b.mx[2] = mx.z; ri_store(t, x, b);
    } else { t.left[x] = childA[x - n]; t.right[x] = childB[x - n]; }
}
// one level of the breadth-first numbering: the children of level d's inner nodes are appended behind the cursor (one atomic per block); bounds[d], bounds[d + 1] delimit level d
__global__ void __launch_bounds__(256) k_ri_bfs(RiTree t, uint* __restrict__ bfs, const uint* __restrict__ bounds, uint d, uint* __restrict__ cursor) {
    __shared__ uint sCnt[4]; __shared__ uint sBase;
    const uint start = bounds[d], end = bounds[d + 1u], wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint base = start + blockIdx.x * 256u; base < end; base += gridDim.x * 256u) {
        const uint i = base + threadIdx.x; uint l = RI_NONE, r = RI_NONE;
        if (i < end) { const uint id = bfs[i]; l = t.left[id]; r = t.right[id]; }
        const bool has = l != RI_NONE;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
        if (lane == 0u) sCnt[wave] = 2u * (uint)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0u) { uint tot = 0; for (uint w = 0; w < 4u; w++) { const uint c = sCnt[w]; sCnt[w] = tot; tot += c; } sBase = tot ? atomicAdd(cursor, tot) : 0u; }
        __syncthreads();
        if (has) { const uint at = sBase + sCnt[wave] + 2u * (uint)__popcll(m & ((1ull << lane) - 1ull)); bfs[at] = l; bfs[at + 1u] = r; }
        __syncthreads();
    }
}
__global__ void k_ri_close(uint* __restrict__ bounds, uint d, const uint* __restrict__ cursor) { if (threadIdx.x == 0u && blockIdx.x == 0u) bounds[d + 2u] = *cursor; }      // level d + 1 ends where the cursor stands
// boxes (and leaf counts) of one level's inner nodes from their children: launched from the deepest level up
__global__ void __launch_bounds__(256) k_ri_refit(RiTree t, const uint* __restrict__ bfs, const uint* __restrict__ bounds, uint d, uint* __restrict__ cnt) {
    const uint start = bounds[d], end = bounds[d + 1u];
    for (uint i = start + blockIdx.x * 256u + threadIdx.x; i < end; i += gridDim.x * 256u) {
        const uint id = bfs[i];
        if (t.left[id] == RI_NONE) { if (cnt) cnt[id] = 1u; continue; }
        ri_refit_node(t, id);
        if (cnt) cnt[id] = cnt[t.left[id]] + cnt[t.right[id]];
    }
}
__global__ void __launch_bounds__(256) k_ri_search(RiDevice r) {
    const uint x = blockIdx.x * 256u + threadIdx.x; if (x >= r.t.N) return;
    uint target, pivot; const float gain = ri_search(r.t, x, 0.0f, target, pivot);
    bool propose = target != RI_NONE;
    if (propose) propose = gain > 1e-6f * ri_area(ri_load(r.t, r.t.par[x]));
    r.gain[x] = gain; r.target[x] = propose ? target : RI_NONE; r.pivot[x] = pivot; r.lock[x] = 0ull; r.moving[x] = 0ull; r.ok[x] = 0u;
}
__global__ void __launch_bounds__(256) k_ri_lock(RiDevice r) {
    const uint x = blockIdx.x * 256u + threadIdx.x; if (x >= r.t.N || r.target[x] == RI_NONE) return;
    const unsigned long long key = ri_key(r.gain[x], x);
    (void)ri_for_links(r.t, x, r.target[x], [&](uint a) { atomicMax(&r.lock[a], key); return true; });
}
__global__ void __launch_bounds__(256) k_ri_check(RiDevice r) {
    const uint x = blockIdx.x * 256u + threadIdx.x; if (x >= r.t.N || r.target[x] == RI_NONE) return;
    const unsigned long long key = ri_key(r.gain[x], x);
    if (ri_for_links(r.t, x, r.target[x], [&](uint a) { return r.lock[a] == key; })) { r.ok[x] = 1u; r.moving[x] = key; }
}
__global__ void __launch_bounds__(256) k_ri_ring(RiDevice r) {
    const uint x = blockIdx.x * 256u + threadIdx.x; if (x >= r.t.N || r.ok[x] != 1u) return;
    if (ri_gives_way(r.t, x, r.target[x], r.pivot[x], r.moving, ri_key(r.gain[x], x))) r.ok[x] = 2u;
}
__global__ void __launch_bounds__(256) k_ri_apply(RiDevice r, uint* __restrict__ applied) {
    const uint x = blockIdx.x * 256u + threadIdx.x;
    const bool go = x < r.t.N && r.ok[x] == 1u;
    if (go) ri_apply(r.t, x, r.target[x]);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(go);
    if (m && (threadIdx.x & 63u) == 0u) atomicAdd(applied, (uint)__popcll(m));
}
// back into the arrays k_ploc_first / k_ploc_finish read: children, parent with the "I am the right child" bit, leaf counts (written by the last refit)
__global__ void __launch_bounds__(256) k_ri_export(RiTree t, uint n, uint* __restrict__ nodeParent, uint* __restrict__ childA, uint* __restrict__ childB) {
    const uint x = blockIdx.x * 256u + threadIdx.x; if (x >= t.N) return;
    if (x >= n) { childA[x - n] = t.left[x]; childB[x - n] = t.right[x]; }
    const uint p = t.par[x];
    nodeParent[x] = (p == RI_NONE) ? 0u : (p | ((t.right[p] == x) ? 0x80000000u : 0u));
}
// the breadth-first numbering of the whole tree; returns the number of levels (reads the level bounds back every 64 levels: a build step)
static hipError_t ri_levels(const RiDevice& r, uint& levels, hipStream_t st) {
    const uint N = r.t.N; const uint first[3] = {0u, 1u, 1u}; const uint root = N - 1u;
    PT_HIP_TRY(hipMemcpyAsync(r.bfs, &root, 4, hipMemcpyHostToDevice, st)); PT_HIP_TRY(hipMemcpyAsync(r.bounds, first, 8, hipMemcpyHostToDevice, st)); PT_HIP_TRY(hipMemcpyAsync(r.cursor, &first[2], 4, hipMemcpyHostToDevice, st));
    std::vector<uint> hb(RI_MAX_LEVELS + 2u);
    for (uint d0 = 0; d0 < RI_MAX_LEVELS; d0 += 64u) {
        for (uint d = d0; d < d0 + 64u; d++) {
            hipLaunchKernelGGL(k_ri_bfs, dim3(1024), dim3(256), 0, st, r.t, r.bfs, r.bounds, d, r.cursor);
            hipLaunchKernelGGL(k_ri_close, dim3(1), dim3(64), 0, st, r.bounds, d, r.cursor);
        }
        PT_HIP_TRY(hipMemcpyAsync(hb.data(), r.bounds, 4 * (size_t)(d0 + 66u), hipMemcpyDeviceToHost, st)); PT_HIP_TRY(hipStreamSynchronize(st));
        for (uint d = d0; d < d0 + 64u; d++) if (hb[d + 1u] == hb[d]) { levels = d; return (hb[d] == N) ? hipSuccess : hipErrorUnknown; }      // an empty level: every node is numbered (or the tree is broken)
    }
    return hipErrorUnknown;
}
static void ri_refit_all(const RiDevice& r, uint levels, uint* cnt, hipStream_t st) {
    for (uint d = levels; d-- > 0u;) hipLaunchKernelGGL(k_ri_refit, dim3(1024), dim3(256), 0, st, r.t, r.bfs, r.bounds, d, cnt);
}
static hipError_t bvh_reinsert(BvhBuildBuffers& b, uint n, uint passes, hipStream_t st) {
    if (!b.riScratch || n < 64u) return hipSuccess;
    const uint N = 2u * n - 1u; char* base = static_cast<char*>(b.riScratch); size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base + off; off += (bytes + 255u) & ~(size_t)255u; return p; };
    RiDevice r;
    r.t.N = N; r.t.par = (uint*)take(4ull * N); r.t.left = (uint*)take(4ull * N); r.t.right = (uint*)take(4ull * N); r.t.box = (float*)take(32ull * N);
    r.gain = (float*)take(4ull * N); r.target = (uint*)take(4ull * N); r.pivot = (uint*)take(4ull * N); r.lock = (unsigned long long*)take(8ull * N); r.moving = (unsigned long long*)take(8ull * N);
    r.ok = (uint*)take(4ull * N); r.bfs = (uint*)take(4ull * N); r.bounds = (uint*)take(4ull * (RI_MAX_LEVELS + 4u)); r.cursor = (uint*)take(256);
    if (off > b.riScratchBytes) return hipErrorInvalidValue;
    const uint g = (N + 255u) / 256u;
    hipLaunchKernelGGL(k_ri_init, dim3(g), dim3(256), 0, st, n, b.plocParent, b.plocChildA, b.plocChildB, b.triWorld, b.primsSorted, r.t);
    uint levels = 0; PT_HIP_TRY(ri_levels(r, levels, st));
    ri_refit_all(r, levels, nullptr, st);
    uint done = 0;
    for (uint pass = 0; pass < passes; pass++) {
        hipLaunchKernelGGL(k_ri_search, dim3(g), dim3(256), 0, st, r);
        hipLaunchKernelGGL(k_ri_lock, dim3(g), dim3(256), 0, st, r);
        hipLaunchKernelGGL(k_ri_check, dim3(g), dim3(256), 0, st, r);
        hipLaunchKernelGGL(k_ri_ring, dim3(g), dim3(256), 0, st, r);
        PT_HIP_TRY(hipMemsetAsync(r.cursor + 1, 0, 4, st));
        hipLaunchKernelGGL(k_ri_apply, dim3(g), dim3(256), 0, st, r, r.cursor + 1);
        PT_HIP_TRY(ri_levels(r, levels, st));                       // (also the check that the tree is still one tree: every node numbered exactly once)
        ri_refit_all(r, levels, (pass + 1u == passes) ? b.plocCnt : nullptr, st);
        done++;
    }
    if (!passes) ri_refit_all(r, levels, b.plocCnt, st);
    hipLaunchKernelGGL(k_ri_export, dim3(g), dim3(256), 0, st, r.t, n, b.plocParent, b.plocChildA, b.plocChildB);
    b.optimiserPasses = done; b.riLevels = levels;
    return hipGetLastError();
}
// "prefer fast trace": the topology comes from the host's binned-SAH build over the world-space triangles k_tri_setup has just written
static hipError_t bvh_sah(BvhBuildBuffers& b, uint n, hipStream_t st) try {
    std::vector<TriRecord> tw(n);
    PT_HIP_TRY(hipMemcpyAsync(tw.data(), b.triWorld, sizeof(TriRecord) * (size_t)n, hipMemcpyDeviceToHost, st));
    PT_HIP_TRY(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<SahTri> st3(n);
    for (uint i = 0; i < n; i++) {
        const TriRecord& t = tw[i]; SahTri& o = st3[i];
        const float* ax[3] = {t.x, t.y, t.z};
        for (int k = 0; k < 3; k++) { const float* a = ax[k]; o.mn[k] = fminf(a[0], fminf(a[1], a[2])); o.mx[k] = fmaxf(a[0], fmaxf(a[1], a[2])); o.c[k] = a[0] + ((a[1] - a[0]) + (a[2] - a[0])) * (1.0f / 3.0f); }
    }
    std::vector<uint> order(n), cl(n), cr(n), rf(n), rl(n), par(n), lp(n), ab(n);
    b.optimiserPasses = bvh_sah_topology(st3.data(), n, SahTopology{order.data(), cl.data(), cr.data(), rf.data(), rl.data(), par.data(), lp.data(), ab.data()}, BVH_MAX_LEAF, 0u);
    PT_HIP_TRY(hipMemcpyAsync(b.absorb, ab.data(), 4 * (size_t)(n - 1u), hipMemcpyHostToDevice, st));
    const size_t inner = 4 * (size_t)(n - 1u);
    PT_HIP_TRY(hipMemcpyAsync(b.primsSorted, order.data(), 4 * (size_t)n, hipMemcpyHostToDevice, st)); PT_HIP_TRY(hipMemcpyAsync(b.leafParent, lp.data(), 4 * (size_t)n, hipMemcpyHostToDevice, st));
    PT_HIP_TRY(hipMemcpyAsync(b.childL, cl.data(), inner, hipMemcpyHostToDevice, st)); PT_HIP_TRY(hipMemcpyAsync(b.childR, cr.data(), inner, hipMemcpyHostToDevice, st));
    PT_HIP_TRY(hipMemcpyAsync(b.rangeFirst, rf.data(), inner, hipMemcpyHostToDevice, st)); PT_HIP_TRY(hipMemcpyAsync(b.rangeLast, rl.data(), inner, hipMemcpyHostToDevice, st));
    PT_HIP_TRY(hipMemcpyAsync(b.parent, par.data(), inner, hipMemcpyHostToDevice, st));
    PT_HIP_TRY(hipStreamSynchronize(st));                       // (the host vectors go out of scope)
    b.hostBuildMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return hipSuccess;
} catch (...) { return hipErrorOutOfMemory; }      // host allocation or thread creation failed: the caller falls back to the device-side builder
hipError_t bvh_build(BvhBuildBuffers& b, const DeviceScene& sc, uint n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint g = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, st, b.sceneBounds);
    hipLaunchKernelGGL(k_tri_setup, dim3(g), dim3(256), 0, st, sc, n, b.triWorld, b.sceneBounds);
    b.hostBuildMs = 0.f; b.optimiserPasses = 0u; b.wideDpPending = 0u; b.wideFlagsValid = 0u; b.bvh2Stale = false;
    if (b.builder == BVH_BUILDER_SAH && n > 1) {
        if (bvh_sah(b, n, st) == hipSuccess) return bvh_bounds_and_emit(b, sc, n, st);
        (void)hipGetLastError(); b.builder = BVH_BUILDER_PLOC;      // no host memory / threads for the fast-trace topology: build it on the device instead
    }
    b.wideDpPending = (b.builder == BVH_BUILDER_PLOC_OPT && n > 1) ? 1u : 0u; b.wideFlagsValid = 0u;
    hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, st, b.triWorld, n, b.sceneBounds, b.keys, b.prims);
    size_t tmp = b.sortTempBytes;
    PT_HIP_TRY(rocprim::radix_sort_pairs(b.sortTemp, tmp, b.keys, b.keysSorted, b.prims, b.primsSorted, (size_t)n, 0, 64, st));
    if (n > 1) {
        if (b.builder == BVH_BUILDER_KARRAS) hipLaunchKernelGGL(k_karras, dim3(g), dim3(256), 0, st, b.keysSorted, (int)n, b.childL, b.childR, b.parent, b.leafParent, b.rangeFirst, b.rangeLast);
        else PT_HIP_TRY(bvh_ploc(b, n, st));
    }
    return bvh_bounds_and_emit(b, sc, n, st);
}
hipError_t bvh_refit(BvhBuildBuffers& b, const DeviceScene& sc, uint n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint g = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, st, b.sceneBounds);
    static const bool fullRefit = getenv("MI355PT_FULL_REFIT") != nullptr;      // developer A/B switch: round 3's refit (k_tri_setup, sparse table, emit, collapse)
    if (b.wideRefitReady && !fullRefit) {
        hipLaunchKernelGGL(k_refit_world, dim3((n + 1023u) / 1024u), dim3(1024), 0, st, sc, b.triSrc, n, b.triSorted, b.sceneBounds);
        bvh_refit_levels(b, st);
        b.bvh2Stale = true;      // only triSorted, nodes8 and the wide boxes follow the pose: triWorld, the BVH2 nodes and the range tables keep the last full build's (pt_build.h)
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_tri_setup, dim3(g), dim3(256), 0, st, sc, n, b.triWorld, b.sceneBounds);
    b.bvh2Stale = false;
    return bvh_bounds_and_emit(b, sc, n, st);
}

} // namespace ptk
