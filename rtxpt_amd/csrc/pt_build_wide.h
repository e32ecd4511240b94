// mi355pt — cost-driven wide-node assignment on the device: which inner nodes of the binary tree a BVH8 node opens (absorbs) and which become wide nodes of their
// own. The dynamic programme of Ylitie, Karras & Laine 2017 ("Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs") exactly as the host builder
// runs it (pt_build_sah.cpp choose_wide_nodes: every wide-node visit and every leaf visit costs its surface area), restated per node so that it runs level by level:
//   wide_dp_node   bottom-up, one call per inner node once both children are done: C[id][i], i = 1..7 = least cost of representing the sub-tree by at most i roots
//                  (a root is a leaf of the wide tree or a wide node of its own), with the decisions packed into one 64-bit word
//   wide_mark_node top-down, one call per inner node once its parent is done: the budget the parent hands down decides whether the node is opened inside the
//                  parent's wide node (absorb = 1) or starts a wide node; writes the state of the two children
// The per-node functions are host + device code: tests/bvh_wide_check.cpp runs them on the CPU against choose_wide_nodes (identical flags expected), pt_build.hip
// runs them over the levels of a breadth-first numbering of the inner nodes (k_wide_levels / k_wide_dp / k_wide_mark).
#pragma once
#if defined(__HIPCC__)
#define PT_WIDE_HD __host__ __device__
#else
#define PT_WIDE_HD
#endif
typedef unsigned int uint;

namespace ptk {

// decisions of one node: kind[i], i = 1..7, two bits each at bit 2 (i - 1) — 0: one wide node of its own (or a leaf of the wide tree), 1: split the budget between the
// children, 2: as with one root less; split[j], j = 2..8, three bits each at bit 14 + 3 (j - 2): the roots handed to the LEFT child when j roots are split
PT_WIDE_HD inline uint wide_kind(unsigned long long d, uint i) { return (uint)(d >> (2u * (i - 1u))) & 3u; }
PT_WIDE_HD inline uint wide_split(unsigned long long d, uint j) { return (uint)(d >> (14u + 3u * (j - 2u))) & 7u; }
PT_WIDE_HD inline float wide_area(const float* mn, const float* mx) { float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2]; return (x < 0.f) ? 0.f : x * y + y * z + z * x; }

static const uint WIDE_LEAF_BIT = 0x80000000u;
// state of a node in the top-down pass: 0 = not reached (inside a leaf of the wide tree), else 0x100 | budget << 1 | startsWideNode
PT_WIDE_HD inline uint wide_state(uint budget, bool wideRoot) { return 0x100u | (budget << 1) | (wideRoot ? 1u : 0u); }

// lmin / lmax / rmin / rmax: the boxes of the node's two children (what k_node_boxes writes; three floats each are read); count: leaves below the node;
// C: 8 floats per inner node (C[8 id + i]); dec: one word per inner node
PT_WIDE_HD inline void wide_dp_node(uint id, uint childL, uint childR, uint count, uint maxLeaf, const float* lmin, const float* lmax, const float* rmin, const float* rmax,
                                    float* C, unsigned long long* dec) {
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = lmin[a] < rmin[a] ? lmin[a] : rmin[a]; mx[a] = lmax[a] > rmax[a] ? lmax[a] : rmax[a]; }
    const float A = wide_area(mn, mx);
    float* c = C + 8u * (unsigned long long)id;
    unsigned long long d = 0ull;
    if (count <= maxLeaf) { for (uint i = 1; i <= 7; i++) c[i] = A; c[0] = 0.f; dec[id] = 0ull; return; }
    const float al = wide_area(lmin, lmax), ar = wide_area(rmin, rmax);
    const bool leafL = (childL & WIDE_LEAF_BIT) != 0u, leafR = (childR & WIDE_LEAF_BIT) != 0u;
    const float* cl = C + 8u * (unsigned long long)(childL & ~WIDE_LEAF_BIT); const float* cr = C + 8u * (unsigned long long)(childR & ~WIDE_LEAF_BIT);
    float D[9];
    for (uint j = 2; j <= 8; j++) {
        D[j] = 3.402823466e+38f;
        for (uint a = 1; a < j; a++) {
            if (a > 7u || j - a > 7u) continue;
            const float v = (leafL ? al : cl[a]) + (leafR ? ar : cr[j - a]);
            if (v < D[j]) { D[j] = v; d = (d & ~(7ull << (14u + 3u * (j - 2u)))) | ((unsigned long long)a << (14u + 3u * (j - 2u))); }
        }
    }
    c[0] = 0.f; c[1] = A + D[8];                                                    // kind[1] = 0
    for (uint i = 2; i <= 7; i++) { c[i] = c[i - 1]; uint k = 2u; if (D[i] < c[i]) { c[i] = D[i]; k = 1u; } d |= (unsigned long long)k << (2u * (i - 1u)); }
    dec[id] = d;
}
// state: what the parent decided for this node (wide_state, the root starts with budget 8 as a wide node); writes absorb[id] and the children's states
PT_WIDE_HD inline void wide_mark_node(uint id, uint state, uint childL, uint childR, uint count, uint maxLeaf, const unsigned long long* dec, uint* absorb, uint* stateOut) {
    if (!state) return;
    uint i = (state >> 1) & 0x7Fu; bool wideRoot = (state & 1u) != 0u;
    const unsigned long long d = dec[id];
    if (!wideRoot) {
        if (count <= maxLeaf) return;                                              // a leaf of the wide tree
        while (wide_kind(d, i) == 2u) i--;
        if (wide_kind(d, i) == 0u) { wideRoot = true; i = 8u; }                    // a root of the forest: a wide node of its own, with the whole budget
        else absorb[id] = 1u;
    }
    const uint a = wide_split(d, i);
    if (!(childL & WIDE_LEAF_BIT)) stateOut[childL] = wide_state(a, false);
    if (!(childR & WIDE_LEAF_BIT)) stateOut[childR] = wide_state(i - a, false);
}

} // namespace ptk
