// mi355pt — insertion-based optimisation of a binary BVH on the device: parallel re-insertion after Meister & Bittner, "Parallel Reinsertion for Bounding
// Volume Hierarchy Optimization" (Eurographics 2018), the data-parallel form of Bittner, Hapala & Havran 2013 that the host builder runs batched
// (pt_build_sah.cpp optimise). One pass = three kernels over all nodes:
//   search  every node x looks for the position where putting it back adds the least surface area: it climbs its ancestors (the pivot), and below every pivot
//           searches the subtree on the other side depth first, pruned by the best gain so far. Removing x (and its parent p) shrinks the ancestors between p
//           and the pivot; inserting x next to a node t grows t's ancestors below the pivot and adds the new parent:
//               gain(t) = area(p) + SUM_{a between p and pivot} [area(a) - area(a without x)] - SUM_{a from t's parent up to below the pivot} [area(a + x) - area(a)] - area(t + x)
//   lock    every node with a positive gain writes (gain, x) with atomicMax into the six nodes whose links its move rewrites (x, sibling, parent, grandparent, target,
//           target's parent). The largest gain wins a node; ties go to the larger node id. No order of execution enters: the result is deterministic. Locking the
//           whole paths up to the pivot as the paper does lets 4 % of the proposed moves through on the 2.8 M-triangle scene (every far move passes the top of the
//           tree); with the six link nodes, a ring test for the moved subtrees (ri_gives_way) and a refit of the whole tree it is ~10 x as many
//   apply   a move that holds its six locks and gives way to no other is carried out by its own thread: p leaves its place (the sibling moves up) and is linked in
//           above t with x as its other child. Disjoint link sets commute. The boxes of all inner nodes are then recomputed level by level, bottom-up.
// No reference is duplicated and the leaf set is unchanged: the hit definition (pt_scene.h tri_box_accepts) is untouched, the closest hit does not depend on the tree.
// All functions are host + device code: tools/bvh_lab and tests/bvh_reinsert_check.cpp run the same code on the CPU, pt_build.hip runs it as k_reinsert_*.
#pragma once
#if defined(__HIPCC__)
#define PT_RI_HD __host__ __device__
#else
#define PT_RI_HD
#endif
typedef unsigned int uint;

namespace ptk {

static const uint RI_NONE = 0xFFFFFFFFu;
static const uint RI_STACK = 48;          // depth-first search stack per thread (entries beyond it are dropped: a lost candidate, never a wrong tree)

struct RiBox { float mn[3], mx[3]; };
PT_RI_HD inline float ri_area(const RiBox& b) { float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2]; return x * y + y * z + z * x; }
PT_RI_HD inline RiBox ri_union(const RiBox& a, const RiBox& b) {
    RiBox r;
    for (int k = 0; k < 3; k++) { r.mn[k] = a.mn[k] < b.mn[k] ? a.mn[k] : b.mn[k]; r.mx[k] = a.mx[k] > b.mx[k] ? a.mx[k] : b.mx[k]; }
    return r;
}
// the tree: nodes 0 .. N - 1 (leaves and inner nodes alike), par[root] = RI_NONE; left / right of a leaf = RI_NONE; boxes as 8 floats per node (mn.xyz, pad, mx.xyz, pad)
struct RiTree { uint* par; uint* left; uint* right; float* box; uint N; };
PT_RI_HD inline RiBox ri_load(const RiTree& t, uint id) { const float* p = t.box + 8ull * id; RiBox b; b.mn[0] = p[0]; b.mn[1] = p[1]; b.mn[2] = p[2]; b.mx[0] = p[4]; b.mx[1] = p[5]; b.mx[2] = p[6]; return b; }
PT_RI_HD inline void ri_store(const RiTree& t, uint id, const RiBox& b) { float* p = t.box + 8ull * id; p[0] = b.mn[0]; p[1] = b.mn[1]; p[2] = b.mn[2]; p[4] = b.mx[0]; p[5] = b.mx[1]; p[6] = b.mx[2]; }

// search: best target for x. Returns the gain (0: stay) and writes the target and the pivot (the lowest common ancestor of the old and the new place).
// minParentArea: nodes whose parent's box is smaller are not candidates (the optimisation concentrates on the large nodes, as the host's does); steps: nodes visited
PT_RI_HD inline float ri_search(const RiTree& t, uint x, float minParentArea, uint& target, uint& pivotOut, uint* steps = nullptr) {
    target = RI_NONE; pivotOut = RI_NONE;
    const uint p = t.par[x]; if (p == RI_NONE) return 0.f;
    if (t.par[p] == RI_NONE) return 0.f;                               // children of the root stay (the root keeps its id)
    const RiBox xb = ri_load(t, x); const float xa = ri_area(xb);
    const float aP = ri_area(ri_load(t, p));
    if (!(aP >= minParentArea)) return 0.f;
    float best = 0.f, dPivot = 0.f;
    uint pivot = p, pathChild = x; RiBox without; bool haveWithout = false;
    uint stackNode[RI_STACK]; float stackInd[RI_STACK];
    uint visited = 0;
    while (pivot != RI_NONE) {
        const uint sib = (t.left[pivot] == pathChild) ? t.right[pivot] : t.left[pivot];
        if (pathChild != x && pathChild != p) {                        // next to the (shrunken) ancestor pathChild itself: a new node between the pivot and pathChild
            const float gain = aP + dPivot - ri_area(ri_union(without, xb));
            if (gain > best) { best = gain; target = pathChild; pivotOut = pivot; }
        }
        uint sp = 0; stackNode[sp] = sib; stackInd[sp] = 0.f; sp++;
        while (sp) {
            sp--; const uint n = stackNode[sp]; const float ind = stackInd[sp];
            if (!(aP + dPivot - ind - xa > best)) continue;            // even a free direct cost cannot beat the best
            visited++;
            const RiBox nb = ri_load(t, n);
            const float direct = ri_area(ri_union(nb, xb));
            const float gain = aP + dPivot - ind - direct;
            if (gain > best && !(pivot == p && n == sib)) { best = gain; target = n; pivotOut = pivot; }      // (the sibling below the own parent is the old place: gain 0)
            const uint l = t.left[n];
            if (l != RI_NONE) {
                const float indChild = ind + direct - ri_area(nb);
                if (aP + dPivot - indChild - xa > best && sp + 2u <= RI_STACK) { stackNode[sp] = l; stackInd[sp] = indChild; sp++; stackNode[sp] = t.right[n]; stackInd[sp] = indChild; sp++; }
            }
        }
        const RiBox sb = ri_load(t, sib);
        without = haveWithout ? ri_union(without, sb) : sb; haveWithout = true;
        if (pivot != p) dPivot += ri_area(ri_load(t, pivot)) - ri_area(without);
        pathChild = pivot; pivot = t.par[pivot];
    }
    if (steps) *steps = visited;
    return best;
}
PT_RI_HD inline unsigned long long ri_key(float gain, uint x) { union { float f; uint u; } c; c.f = gain; return ((unsigned long long)c.u << 32) | x; }
// the six nodes whose links a move rewrites: x, its sibling s, its parent p, the grandparent g, the target and the target's parent. Two moves with disjoint sets
// rewrite disjoint links. f(node) returns false to stop early.
template <class F> PT_RI_HD inline bool ri_for_links(const RiTree& t, uint x, uint target, F f) {
    const uint p = t.par[x]; const uint s = (t.left[p] == x) ? t.right[p] : t.left[p]; const uint g = t.par[p], tp = t.par[target];
    if (g == RI_NONE || tp == RI_NONE) return false;
    return f(x) && f(s) && f(p) && f(g) && f(target) && f(tp);
}
// Cycles. A move carries the subtree of x to a place below the pivot; were that place inside a subtree that another move carries away, and that move's place in
// turn inside this one (or a longer ring), the moved subtrees would end up hanging on one another. moving[a] = key of the accepted move whose x is a (0: none):
// a move whose way from the target up to the pivot meets a moving node of HIGHER key gives way. In any ring the move that points at the ring's highest key is
// dropped, which opens the ring; moves that merely sit inside a subtree that is carried along are unaffected.
PT_RI_HD inline bool ri_gives_way(const RiTree& t, uint x, uint target, uint pivot, const unsigned long long* moving, unsigned long long key) {
    for (uint a = target; a != pivot && a != RI_NONE; a = t.par[a]) if (a != x && moving[a] > key) return true;
    return false;
}
// apply: re-link only (the boxes are recomputed for the whole tree afterwards: the paths of different moves overlap)
PT_RI_HD inline void ri_apply(const RiTree& t, uint x, uint target) {
    const uint p = t.par[x], g = t.par[p]; const uint s = (t.left[p] == x) ? t.right[p] : t.left[p];
    if (t.left[g] == p) t.left[g] = s; else t.right[g] = s;
    t.par[s] = g;
    const uint tp = t.par[target];
    if (t.left[tp] == target) t.left[tp] = p; else t.right[tp] = p;
    t.par[p] = tp; t.left[p] = target; t.right[p] = x; t.par[target] = p; t.par[x] = p;
}
PT_RI_HD inline void ri_refit_node(const RiTree& t, uint a) { ri_store(t, a, ri_union(ri_load(t, t.left[a]), ri_load(t, t.right[a]))); }

} // namespace ptk
