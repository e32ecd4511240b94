// developer tool (CPU only, not part of the product or the oracle): how much would a better tree buy the BVH8 traversal?
// Builds the scene's BVH with several builders (Morton LBVH as pt_build.hip does, PLOC, binned top-down SAH), collapses each to BVH8 with the
// product's greedy rule or a cost-driven rule, and counts node visits / leaf visits / triangle tests per ray of a nearest-first traversal with the
// product's pruning rules over a set of path-like rays (camera rays + cosine-distributed bounces). Input: tools/bvh_lab/dump_tris.py.
//   g++ -O3 -march=native -fopenmp -std=c++17 tools/bvh_lab/bvh_lab.cpp rtxpt_amd/csrc/pt_build_sah.cpp -lpthread -o /tmp/bvh_lab && /tmp/bvh_lab /tmp/bvh_lab_tris.bin
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <vector>
#include <omp.h>
#include "../../rtxpt_amd/csrc/pt_build_sah.h"      // LAB_PRODUCT: the product's host builder (link rtxpt_amd/csrc/pt_build_sah.cpp)
#include "../../rtxpt_amd/csrc/pt_build_reinsert.h" // LAB_PARALLEL_REINS: the product's device-side optimiser, run on the CPU

struct V3 { float x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline V3 normalize(V3 a) { float l = std::sqrt(dot(a, a)); return a * (1.0f / l); }
struct Box { V3 mn{3e38f, 3e38f, 3e38f}, mx{-3e38f, -3e38f, -3e38f};
    void grow(V3 p) { mn = vmin(mn, p); mx = vmax(mx, p); }
    void grow(const Box& b) { mn = vmin(mn, b.mn); mx = vmax(mx, b.mx); }
    float area() const { V3 e = mx - mn; return e.x * e.y + e.y * e.z + e.z * e.x; } };
static inline Box unite(const Box& a, const Box& b) { Box r = a; r.grow(b); return r; }
struct Tri { V3 v0, e1, e2; };

struct Node { Box box; int l = -1, r = -1; int first = 0, count = 0; };      // count > 0: leaf over order[first .. first+count)
struct Bvh2 { std::vector<Node> nodes; std::vector<int> order; int root = 0; };

static std::vector<Tri> tris; static std::vector<Box> tbox; static std::vector<V3> tcen;

static uint64_t expand21(uint32_t v) { uint64_t x = v & 0x1FFFFF; x = (x | x << 32) & 0x1F00000000FFFFull; x = (x | x << 16) & 0x1F0000FF0000FFull; x = (x | x << 8) & 0x100F00F00F00F00Full; x = (x | x << 4) & 0x10C30C30C30C30C3ull; x = (x | x << 2) & 0x1249249249249249ull; return x; }
static std::vector<int> morton_order(std::vector<uint64_t>* keysOut = nullptr) {
    Box sb; for (auto& b : tbox) sb.grow(b);
    V3 ext = sb.mx - sb.mn; int n = (int)tris.size();
    std::vector<uint64_t> keys(n);
    for (int i = 0; i < n; i++) {
        V3 c = tcen[i];
        auto q = [](float s) { return (uint32_t)std::min(std::max(s * 2097152.0f, 0.0f), 2097151.0f); };
        keys[i] = (expand21(q((c.x - sb.mn.x) / ext.x)) << 2) | (expand21(q((c.y - sb.mn.y) / ext.y)) << 1) | expand21(q((c.z - sb.mn.z) / ext.z));
    }
    std::vector<int> ord(n); std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return keys[a] < keys[b]; });
    if (keysOut) { keysOut->resize(n); for (int i = 0; i < n; i++) (*keysOut)[i] = keys[ord[i]]; }
    return ord;
}

// ---- LBVH (same topology as Karras over sorted 63-bit codes: split at the highest differing bit; ties split on the index)
static Bvh2 build_lbvh() {
    Bvh2 b; std::vector<uint64_t> keys; b.order = morton_order(&keys); int n = (int)tris.size();
    b.nodes.reserve(2 * n);
    auto delta = [&](int i, int j) -> int { if (keys[i] == keys[j]) return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j); return __builtin_clzll(keys[i] ^ keys[j]); };
    std::function<int(int, int)> rec = [&](int lo, int hi) -> int {
        int id = (int)b.nodes.size(); b.nodes.emplace_back();
        if (lo == hi) { Node& nd = b.nodes[id]; nd.first = lo; nd.count = 1; nd.box = tbox[b.order[lo]]; return id; }
        int d = delta(lo, hi), s = lo, step = hi - lo;       // largest s in [lo, hi) with delta(lo, s) > d
        do { step = (step + 1) >> 1; int ns = s + step; if (ns < hi && delta(lo, ns) > d) s = ns; } while (step > 1);
        int l = rec(lo, s), r = rec(s + 1, hi);
        Node& nd = b.nodes[id]; nd.l = l; nd.r = r; nd.box = unite(b.nodes[l].box, b.nodes[r].box); nd.first = lo; nd.count = 0;
        return id;
    };
    b.root = rec(0, n - 1);
    return b;
}

// ---- PLOC (Meister & Bittner 2018): nearest-neighbour merging of Morton-ordered clusters within a window of `radius`
static Bvh2 build_ploc(int radius) {
    Bvh2 b; std::vector<int> mord = morton_order(); int n = (int)tris.size();
    b.nodes.resize(n); b.nodes.reserve(2 * n);
    std::vector<int> cl(n), nn(n), nxt; cl.reserve(n);
    for (int i = 0; i < n; i++) { Node& nd = b.nodes[i]; nd.first = mord[i]; nd.count = 1; nd.box = tbox[mord[i]]; cl[i] = i; }      // first = triangle id for now
    int m = n, iters = 0;
    while (m > 1) {
        #pragma omp parallel for schedule(static)
        for (int i = 0; i < m; i++) {
            float best = 3e38f; int bj = -1; const Box& bi = b.nodes[cl[i]].box;
            for (int j = std::max(0, i - radius); j <= std::min(m - 1, i + radius); j++) if (j != i) { float a = unite(bi, b.nodes[cl[j]].box).area(); if (a < best) { best = a; bj = j; } }
            nn[i] = bj;
        }
        nxt.clear();
        for (int i = 0; i < m; i++) {
            int j = nn[i];
            if (nn[j] == i) { if (i < j) { int id = (int)b.nodes.size(); b.nodes.emplace_back(); Node& nd = b.nodes[id]; nd.l = cl[i]; nd.r = cl[j]; nd.box = unite(b.nodes[cl[i]].box, b.nodes[cl[j]].box); nxt.push_back(id); } }
            else nxt.push_back(cl[i]);
        }
        cl.swap(nxt); m = (int)cl.size(); iters++;
    }
    b.root = cl[0];
    // DFS order of the leaves -> contiguous ranges
    b.order.clear(); b.order.reserve(n);
    std::vector<int> st{b.root}; std::vector<int> post;
    std::function<void(int)> dfs = [&](int id) { Node& nd = b.nodes[id]; if (nd.count) { int t = nd.first; nd.first = (int)b.order.size(); b.order.push_back(t); return; } int f = (int)b.order.size(); dfs(nd.l); dfs(nd.r); nd.first = f; };
    // iterative to avoid deep recursion
    { struct F { int id, stage; }; std::vector<F> s{{b.root, 0}};
      while (!s.empty()) { F f = s.back(); s.pop_back(); Node& nd = b.nodes[f.id];
        if (nd.count) { int t = nd.first; nd.first = (int)b.order.size(); b.order.push_back(t); continue; }
        if (f.stage == 0) { nd.first = (int)b.order.size(); s.push_back({nd.r, 0}); s.push_back({nd.l, 0}); } } }
    fprintf(stderr, "  ploc r=%d: %d iterations\n", radius, iters);
    return b;
}

// ---- binned top-down SAH (object splits), to single-triangle leaves
static Bvh2 build_sah(int bins) {
    Bvh2 b; int n = (int)tris.size(); b.order.resize(n); std::iota(b.order.begin(), b.order.end(), 0); b.nodes.reserve(2 * n);
    struct Job { int id, lo, hi; };
    b.nodes.emplace_back(); std::vector<Job> jobs{{0, 0, n}};
    while (!jobs.empty()) {
        Job j = jobs.back(); jobs.pop_back();
        Box nb, cb; for (int i = j.lo; i < j.hi; i++) { nb.grow(tbox[b.order[i]]); cb.grow(tcen[b.order[i]]); }
        b.nodes[j.id].box = nb; b.nodes[j.id].first = j.lo;
        int cnt = j.hi - j.lo;
        if (cnt == 1) { b.nodes[j.id].count = 1; continue; }
        float bestCost = 3e38f; int bestAxis = -1, bestBin = -1;
        for (int ax = 0; ax < 3; ax++) {
            float lo = (&cb.mn.x)[ax], hi = (&cb.mx.x)[ax]; if (!(hi > lo)) continue;
            std::vector<Box> bb(bins); std::vector<int> bc(bins, 0); float k = bins / (hi - lo);
            for (int i = j.lo; i < j.hi; i++) { int t = b.order[i]; int bi = std::min(bins - 1, (int)(((&tcen[t].x)[ax] - lo) * k)); bb[bi].grow(tbox[t]); bc[bi]++; }
            std::vector<float> ra(bins); Box acc; int c = 0;
            for (int i = bins - 1; i > 0; i--) { acc.grow(bb[i]); c += bc[i]; ra[i] = c ? acc.area() * c : 3e38f; }
            acc = Box(); c = 0;
            for (int i = 0; i < bins - 1; i++) { acc.grow(bb[i]); c += bc[i]; if (!c || c == cnt) continue; float cost = acc.area() * c + ra[i + 1]; if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = i; } }
        }
        int mid;
        if (bestAxis < 0) mid = j.lo + cnt / 2;
        else { float lo = (&cb.mn.x)[bestAxis], hi = (&cb.mx.x)[bestAxis], k = bins / (hi - lo);
               mid = (int)(std::partition(b.order.begin() + j.lo, b.order.begin() + j.hi, [&](int t) { return std::min(bins - 1, (int)(((&tcen[t].x)[bestAxis] - lo) * k)) <= bestBin; }) - b.order.begin());
               if (mid == j.lo || mid == j.hi) mid = j.lo + cnt / 2; }
        int l = (int)b.nodes.size(); b.nodes.emplace_back(); int r = (int)b.nodes.size(); b.nodes.emplace_back();
        b.nodes[j.id].l = l; b.nodes[j.id].r = r;
        jobs.push_back({l, j.lo, mid}); jobs.push_back({r, mid, j.hi});
    }
    b.root = 0;
    return b;
}


// ---- SBVH (Stich et al. 2009): binned object splits + chopped-binning spatial splits with reference unsplitting, to single-reference leaves.
// alpha: spatial splits are tried when the object split's children overlap by more than alpha x the root area; budget: at most `budget` x n references.
static Box isect_box(const Box& a, const Box& b) { Box r; r.mn = vmax(a.mn, b.mn); r.mx = vmin(a.mx, b.mx); return r; }
static inline bool box_valid(const Box& b) { return b.mn.x <= b.mx.x && b.mn.y <= b.mx.y && b.mn.z <= b.mx.z; }
static Box clip_tri(int t, int ax, float lo, float hi, const Box& cur) {
    V3 a = tris[t].v0, poly[2][10]; poly[0][0] = a; poly[0][1] = a + tris[t].e1; poly[0][2] = a + tris[t].e2; int np = 3, src = 0;
    for (int pass = 0; pass < 2; pass++) {
        float plane = pass ? hi : lo, sgn = pass ? -1.f : 1.f; V3* in = poly[src]; V3* out = poly[src ^ 1]; int no = 0;
        for (int i = 0; i < np; i++) {
            V3 p = in[i], q = in[(i + 1) % np]; float dp = sgn * ((&p.x)[ax] - plane), dq = sgn * ((&q.x)[ax] - plane);
            if (dp >= 0) out[no++] = p;
            if ((dp > 0 && dq < 0) || (dp < 0 && dq > 0)) { float u = dp / (dp - dq); V3 x = p + (q - p) * u; (&x.x)[ax] = plane; out[no++] = x; }
        }
        np = no; src ^= 1; if (!np) break;
    }
    Box r; for (int i = 0; i < np; i++) r.grow(poly[src][i]);
    return isect_box(r, cur);
}
struct Ref { Box b; int tri; };
static Bvh2 build_sbvh(int bins, float alpha, float budget) {
    Bvh2 b; int n = (int)tris.size(); b.nodes.reserve(3 * n); b.order.reserve((size_t)(n * budget) + 16);
    std::vector<Ref> all(n); Box rootBox; for (int i = 0; i < n; i++) { all[i] = {tbox[i], i}; rootBox.grow(tbox[i]); }
    const float rootArea = rootBox.area(); size_t totalRefs = n, maxRefs = (size_t)(n * budget); size_t nSpatial = 0;
    struct Job { int id; std::vector<Ref> refs; };
    b.nodes.emplace_back(); std::vector<Job> jobs; jobs.push_back({0, std::move(all)});
    while (!jobs.empty()) {
        Job j = std::move(jobs.back()); jobs.pop_back();
        std::vector<Ref>& R = j.refs; int cnt = (int)R.size();
        Box nb, cb; for (auto& r : R) { nb.grow(r.b); cb.grow((r.b.mn + r.b.mx) * 0.5f); }
        b.nodes[j.id].box = nb;
        if (cnt == 1) { b.nodes[j.id].first = (int)b.order.size(); b.nodes[j.id].count = 1; b.order.push_back(R[0].tri); continue; }
        // object split
        float objCost = 3e38f; int objAxis = -1, objBin = -1; Box objL, objR;
        for (int ax = 0; ax < 3; ax++) {
            float lo = (&cb.mn.x)[ax], hi = (&cb.mx.x)[ax]; if (!(hi > lo)) continue;
            std::vector<Box> bb(bins); std::vector<int> bc(bins, 0); float k = bins / (hi - lo);
            for (auto& r : R) { int bi = std::min(bins - 1, (int)((((&r.b.mn.x)[ax] + (&r.b.mx.x)[ax]) * 0.5f - lo) * k)); bb[bi].grow(r.b); bc[bi]++; }
            std::vector<float> ra(bins); std::vector<Box> rb(bins); Box acc; int c = 0;
            for (int i = bins - 1; i > 0; i--) { acc.grow(bb[i]); c += bc[i]; ra[i] = c ? acc.area() * c : 3e38f; rb[i] = acc; }
            acc = Box(); c = 0;
            for (int i = 0; i < bins - 1; i++) { acc.grow(bb[i]); c += bc[i]; if (!c || c == cnt) continue; float cost = acc.area() * c + ra[i + 1]; if (cost < objCost) { objCost = cost; objAxis = ax; objBin = i; objL = acc; objR = rb[i + 1]; } }
        }
        // spatial split
        float spCost = 3e38f; int spAxis = -1; float spPos = 0;
        bool trySpatial = totalRefs < maxRefs && cnt > 1;
        if (trySpatial && objAxis >= 0) { Box ov = isect_box(objL, objR); trySpatial = box_valid(ov) && ov.area() / rootArea > alpha; }
        if (trySpatial) {
            for (int ax = 0; ax < 3; ax++) {
                float lo = (&nb.mn.x)[ax], hi = (&nb.mx.x)[ax]; if (!(hi > lo)) continue;
                std::vector<Box> bb(bins); std::vector<int> en(bins, 0), ex(bins, 0); float k = bins / (hi - lo), w = (hi - lo) / bins;
                for (auto& r : R) {
                    int b0 = std::min(bins - 1, std::max(0, (int)(((&r.b.mn.x)[ax] - lo) * k))), b1 = std::min(bins - 1, std::max(b0, (int)(((&r.b.mx.x)[ax] - lo) * k)));
                    en[b0]++; ex[b1]++;
                    if (b0 == b1) { bb[b0].grow(r.b); continue; }
                    for (int bi = b0; bi <= b1; bi++) { Box c = clip_tri(r.tri, ax, lo + bi * w, bi == bins - 1 ? hi : lo + (bi + 1) * w, r.b); if (box_valid(c)) bb[bi].grow(c); }
                }
                std::vector<float> ra(bins); Box acc; int c = 0;
                for (int i = bins - 1; i > 0; i--) { acc.grow(bb[i]); c += ex[i]; ra[i] = c ? acc.area() * c : 3e38f; }
                acc = Box(); c = 0;
                for (int i = 0; i < bins - 1; i++) { acc.grow(bb[i]); c += en[i]; if (!c) continue; float cost = acc.area() * c + ra[i + 1]; if (ra[i + 1] < 3e38f && cost < spCost) { spCost = cost; spAxis = ax; spPos = lo + (i + 1) * w; } }
            }
        }
        std::vector<Ref> L, Rr;
        if (spAxis >= 0 && spCost < objCost) {
            // partition with reference unsplitting
            Box lb, rbx; int nl = 0, nr = 0; std::vector<int> straddle;
            for (int i = 0; i < cnt; i++) { const Ref& r = R[i];
                if ((&r.b.mx.x)[spAxis] <= spPos) { lb.grow(r.b); nl++; L.push_back(r); }
                else if ((&r.b.mn.x)[spAxis] >= spPos) { rbx.grow(r.b); nr++; Rr.push_back(r); }
                else straddle.push_back(i); }
            for (int i : straddle) { const Ref& r = R[i];
                Box cl = clip_tri(r.tri, spAxis, (&nb.mn.x)[spAxis], spPos, r.b), cr = clip_tri(r.tri, spAxis, spPos, (&nb.mx.x)[spAxis], r.b);
                bool vl = box_valid(cl), vr = box_valid(cr);
                if (!vl && !vr) { L.push_back(r); lb.grow(r.b); nl++; continue; }
                if (!vl) { Rr.push_back({cr, r.tri}); rbx.grow(cr); nr++; continue; }
                if (!vr) { L.push_back({cl, r.tri}); lb.grow(cl); nl++; continue; }
                Box lbs = unite(lb, cl), rbs = unite(rbx, cr), lbu = unite(lb, r.b), rbu = unite(rbx, r.b);
                float cSplit = lbs.area() * (nl + 1) + rbs.area() * (nr + 1), cLeft = lbu.area() * (nl + 1) + rbx.area() * nr, cRight = lb.area() * nl + rbu.area() * (nr + 1);
                if (nl == 0 && !box_valid(lb)) cRight = 3e38f; if (nr == 0 && !box_valid(rbx)) cLeft = 3e38f;
                if (cSplit <= cLeft && cSplit <= cRight) { L.push_back({cl, r.tri}); Rr.push_back({cr, r.tri}); lb = lbs; rbx = rbs; nl++; nr++; totalRefs++; }
                else if (cLeft <= cRight) { L.push_back(r); lb = lbu; nl++; }
                else { Rr.push_back(r); rbx = rbu; nr++; } }
            if (L.empty() || Rr.empty() || ((int)L.size() == cnt && (int)Rr.size() == cnt)) { L.clear(); Rr.clear(); }
            else nSpatial++;
        }
        if (L.empty()) {
            if (objAxis >= 0) { float lo = (&cb.mn.x)[objAxis], hi = (&cb.mx.x)[objAxis], k = bins / (hi - lo);
                for (auto& r : R) { int bi = std::min(bins - 1, (int)((((&r.b.mn.x)[objAxis] + (&r.b.mx.x)[objAxis]) * 0.5f - lo) * k)); (bi <= objBin ? L : Rr).push_back(r); } }
            if (L.empty() || Rr.empty()) { L.assign(R.begin(), R.begin() + cnt / 2); Rr.assign(R.begin() + cnt / 2, R.end()); }
        }
        { std::vector<Ref>().swap(R); }
        int l = (int)b.nodes.size(); b.nodes.emplace_back(); int r = (int)b.nodes.size(); b.nodes.emplace_back();
        b.nodes[j.id].l = l; b.nodes[j.id].r = r;
        jobs.push_back({r, std::move(Rr)}); jobs.push_back({l, std::move(L)});
    }
    // leaves were appended in DFS order (left first): ranges are contiguous; fix `first` of inner nodes
    { std::vector<int> st{0}, po; while (!st.empty()) { int id = st.back(); st.pop_back(); po.push_back(id); if (!b.nodes[id].count) { st.push_back(b.nodes[id].l); st.push_back(b.nodes[id].r); } }
      for (int k = (int)po.size() - 1; k >= 0; k--) { Node& nd = b.nodes[po[k]]; if (!nd.count) nd.first = b.nodes[nd.l].first; } }
    b.root = 0;
    fprintf(stderr, "  sbvh alpha %g: %zu references for %d triangles (x%.3f), %zu spatial splits\n", alpha, b.order.size(), n, (double)b.order.size() / n, nSpatial);
    return b;
}

static int subtree_tris(const Bvh2& b, int id, std::vector<int>& cnt) { const Node& nd = b.nodes[id]; if (nd.count) return cnt[id] = nd.count; return cnt[id] = subtree_tris(b, nd.l, cnt) + subtree_tris(b, nd.r, cnt); }
static double sah_cost2(const Bvh2& b) { double c = 0; double ra = b.nodes[b.root].box.area(); for (auto& nd : b.nodes) c += nd.box.area() / ra * (nd.count ? nd.count : 1.0); return c; }


// ---- insertion-based optimisation (Bittner et al. 2013; the search of Meister & Bittner 2018): take a node out of the tree, look for the position where
// putting it back costs the least surface area (branch and bound from the root), put it there. No duplicated references, the leaf set is unchanged.
static void renumber_dfs(Bvh2& b) {
    std::vector<int> ord; ord.reserve(b.order.size());
    std::vector<int> st{b.root}; std::vector<int> po;
    while (!st.empty()) { int id = st.back(); st.pop_back(); po.push_back(id); Node& nd = b.nodes[id]; if (nd.count) { int t = b.order[nd.first]; nd.first = (int)ord.size(); ord.push_back(t); } else { st.push_back(nd.r); st.push_back(nd.l); } }
    for (int k = (int)po.size() - 1; k >= 0; k--) { Node& nd = b.nodes[po[k]]; if (!nd.count) nd.first = b.nodes[nd.l].first; }
    b.order.swap(ord);
}
static void optimize_reinsert(Bvh2& b, int passes, float fraction) {
    int N = (int)b.nodes.size(); std::vector<int> parent(N, -1);
    for (int i = 0; i < N; i++) if (!b.nodes[i].count) { parent[b.nodes[i].l] = i; parent[b.nodes[i].r] = i; }
    auto refit_up = [&](int a) { while (a >= 0) { Node& nd = b.nodes[a]; Box nb = unite(b.nodes[nd.l].box, b.nodes[nd.r].box); nd.box = nb; a = parent[a]; } };
    uint32_t rs = 99u; size_t moved = 0;
    for (int pass = 0; pass < passes; pass++) {
        // candidates: the nodes whose parent's box is much larger than their own (area of parent - area), top `fraction`
        std::vector<std::pair<float, int>> cand; cand.reserve(N);
        for (int i = 0; i < N; i++) { int p = parent[i]; if (p < 0 || parent[p] < 0) continue; cand.push_back({b.nodes[p].box.area() - b.nodes[i].box.area() * 0.0f + b.nodes[p].box.area() * 0.f + (b.nodes[p].box.area()), i}); }
        size_t take = (size_t)(cand.size() * fraction); if (take < 1) take = 1;
        std::partial_sort(cand.begin(), cand.begin() + take, cand.end(), [](auto& a, auto& c) { return a.first > c.first; });
        for (size_t ci = 0; ci < take; ci++) {
            int x = cand[ci].second, p = parent[x]; if (p < 0) continue; int g = parent[p]; if (g < 0) continue;
            int s = b.nodes[p].l == x ? b.nodes[p].r : b.nodes[p].l;
            // remove x and p: s takes p's place under g
            (b.nodes[g].l == p ? b.nodes[g].l : b.nodes[g].r) = s; parent[s] = g; refit_up(g);
            const Box xb = b.nodes[x].box; const float xa = xb.area();
            // branch and bound from the root
            struct Q { float induced; int id; }; auto cmp = [](const Q& a, const Q& c) { return a.induced > c.induced; };
            std::vector<Q> heap; heap.push_back({0.f, b.root}); float best = 3e38f; int bestNode = -1;
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end(), cmp); Q q = heap.back(); heap.pop_back();
                if (q.induced + xa >= best) break;
                const Node& nd = b.nodes[q.id]; float direct = unite(nd.box, xb).area(); float tot = q.induced + direct;
                if (tot < best) { best = tot; bestNode = q.id; }
                float ind = tot - nd.box.area();
                if (!nd.count && ind + xa < best) { heap.push_back({ind, nd.l}); std::push_heap(heap.begin(), heap.end(), cmp); heap.push_back({ind, nd.r}); std::push_heap(heap.begin(), heap.end(), cmp); }
            }
            // insert: p becomes the parent of (bestNode, x) where bestNode was
            int n = bestNode, np = parent[n];
            if (np < 0) { b.root = p; parent[p] = -1; } else { (b.nodes[np].l == n ? b.nodes[np].l : b.nodes[np].r) = p; parent[p] = np; }
            b.nodes[p].l = n; b.nodes[p].r = x; parent[n] = p; parent[x] = p; refit_up(p);
            moved += n != s;
        }
        fprintf(stderr, "  reinsert pass %d: %zu candidates, %zu moved so far, sah %.2f\n", pass, take, moved, sah_cost2(b));
    }
    (void)rs;
    renumber_dfs(b);
}


// ---- the device-side optimiser (pt_build_reinsert.h: search / lock / check / apply over all nodes) run pass by pass on the CPU, exactly as k_reinsert_* run it
static void optimize_parallel_reinsert(Bvh2& b, int passes, float fraction) {
    const uint N = (uint)b.nodes.size();
    std::vector<uint> par(N, ptk::RI_NONE), left(N, ptk::RI_NONE), right(N, ptk::RI_NONE); std::vector<float> box(8 * (size_t)N, 0.f);
    for (uint i = 0; i < N; i++) { const Node& nd = b.nodes[i]; if (!nd.count) { left[i] = (uint)nd.l; right[i] = (uint)nd.r; par[nd.l] = i; par[nd.r] = i; }
        float* q = &box[8 * (size_t)i]; q[0] = nd.box.mn.x; q[1] = nd.box.mn.y; q[2] = nd.box.mn.z; q[4] = nd.box.mx.x; q[5] = nd.box.mx.y; q[6] = nd.box.mx.z; }
    ptk::RiTree t{par.data(), left.data(), right.data(), box.data(), N};
    std::vector<float> gain(N); std::vector<uint> target(N), pivot(N); std::vector<unsigned long long> lock(N); std::vector<unsigned char> ok(N);
    for (int pass = 0; pass < passes; pass++) {
        float thr = 0.f;
        if (fraction < 1.f) { std::vector<float> pa; pa.reserve(N); for (uint i = 0; i < N; i++) { uint p = par[i]; if (p != ptk::RI_NONE && par[p] != ptk::RI_NONE) pa.push_back(ptk::ri_area(ptk::ri_load(t, p))); }
            size_t k = (size_t)(pa.size() * fraction); if (k < 1) k = 1; std::nth_element(pa.begin(), pa.begin() + (k - 1), pa.end(), [](float a, float c) { return a > c; }); thr = pa[k - 1]; }
        unsigned long long steps = 0, maxSteps = 0; size_t proposed = 0, applied = 0;
        #pragma omp parallel for schedule(dynamic, 4096) reduction(+:steps, proposed) reduction(max:maxSteps)
        for (uint x = 0; x < N; x++) { uint st = 0; gain[x] = ptk::ri_search(t, x, thr, target[x], pivot[x], &st); steps += st; if (st > maxSteps) maxSteps = st;
            if (!(gain[x] > 1e-6f * ptk::ri_area(ptk::ri_load(t, par[x] == ptk::RI_NONE ? x : par[x])))) target[x] = ptk::RI_NONE; else proposed++; }
        std::fill(lock.begin(), lock.end(), 0ull);
        for (uint x = 0; x < N; x++) if (target[x] != ptk::RI_NONE) { const unsigned long long key = ptk::ri_key(gain[x], x); ptk::ri_for_links(t, x, target[x], [&](uint a) { if (lock[a] < key) lock[a] = key; return true; }); }
        std::vector<unsigned long long> moving(N, 0ull);
        for (uint x = 0; x < N; x++) { ok[x] = 0; if (target[x] != ptk::RI_NONE) { const unsigned long long key = ptk::ri_key(gain[x], x); ok[x] = ptk::ri_for_links(t, x, target[x], [&](uint a) { return lock[a] == key; }) ? 1 : 0; if (ok[x]) moving[x] = key; } }
        size_t gaveWay = 0;
        for (uint x = 0; x < N; x++) if (ok[x] && ptk::ri_gives_way(t, x, target[x], pivot[x], moving.data(), ptk::ri_key(gain[x], x))) { ok[x] = 2; gaveWay++; }
        for (uint x = 0; x < N; x++) if (ok[x] == 1) { ptk::ri_apply(t, x, target[x]); applied++; }
        { // refit everything bottom-up (levels of a breadth-first numbering, deepest first) and check that the tree is still one tree
          std::vector<uint> bfs{(uint)b.root}; for (size_t k = 0; k < bfs.size(); k++) { uint id = bfs[k]; if (left[id] != ptk::RI_NONE) { bfs.push_back(left[id]); bfs.push_back(right[id]); } }
          if (bfs.size() != N) { fprintf(stderr, "TREE BROKEN: %zu of %u nodes reachable\n", bfs.size(), N); exit(1); }
          for (size_t k = bfs.size(); k-- > 0;) if (left[bfs[k]] != ptk::RI_NONE) ptk::ri_refit_node(t, bfs[k]); }
        (void)gaveWay;
        double c = 0; for (uint i = 0; i < N; i++) if (left[i] != ptk::RI_NONE) c += ptk::ri_area(ptk::ri_load(t, i));
        fprintf(stderr, "  parallel reinsertion pass %d: %zu proposed, %zu applied, %.1f search steps per node (max %llu), inner area / root area %.2f\n", pass, proposed, applied, (double)steps / N, maxSteps, c / ptk::ri_area(ptk::ri_load(t, (uint)b.root)));
    }
    for (uint i = 0; i < N; i++) { Node& nd = b.nodes[i]; if (!nd.count) { nd.l = (int)left[i]; nd.r = (int)right[i]; } const float* q = &box[8 * (size_t)i]; nd.box.mn = {q[0], q[1], q[2]}; nd.box.mx = {q[4], q[5], q[6]}; }
    renumber_dfs(b);
}

// the product's "prefer fast trace" topology (pt_build_sah.cpp) as a lab tree
static Bvh2 build_product() {
    int n = (int)tris.size(); std::vector<ptk::SahTri> t(n);
    for (int i = 0; i < n; i++) { t[i] = {{tbox[i].mn.x, tbox[i].mn.y, tbox[i].mn.z}, {tbox[i].mx.x, tbox[i].mx.y, tbox[i].mx.z}, {tcen[i].x, tcen[i].y, tcen[i].z}}; }
    std::vector<uint> order(n), cl(n), cr(n), rf(n), rl(n), par(n), lp(n), ab(n);
    double t0 = omp_get_wtime();
    ptk::bvh_sah_topology(t.data(), (uint)n, ptk::SahTopology{order.data(), cl.data(), cr.data(), rf.data(), rl.data(), par.data(), lp.data(), ab.data()}, 4u, 0u);
    fprintf(stderr, "  product builder: %.3f s\n", omp_get_wtime() - t0);
    Bvh2 b; b.order.assign(order.begin(), order.end()); b.nodes.resize(2 * (size_t)n - 1);
    for (int q = 0; q < n; q++) { Node& nd = b.nodes[n - 1 + q]; nd.first = q; nd.count = 1; nd.box = tbox[order[q]]; }
    auto id = [&](uint ref) { return (ref >> 31) ? (int)(n - 1 + (ref & 0x7FFFFFFFu)) : (int)ref; };
    for (int i = 0; i < n - 1; i++) { Node& nd = b.nodes[i]; nd.l = id(cl[i]); nd.r = id(cr[i]); nd.first = (int)rf[i]; nd.count = 0; }
    std::vector<int> st{0}, po; while (!st.empty()) { int k = st.back(); st.pop_back(); po.push_back(k); if (!b.nodes[k].count) { st.push_back(b.nodes[k].l); st.push_back(b.nodes[k].r); } }
    for (int k = (int)po.size() - 1; k >= 0; k--) { Node& nd = b.nodes[po[k]]; if (!nd.count) nd.box = unite(b.nodes[nd.l].box, b.nodes[nd.r].box); }
    b.root = 0; return b;
}

// ---- BVH8
struct Wide { Box cb[8]; int ref[8]; int n = 0; };     // ref >= 0: wide node; ref < 0: leaf ~ref = first<<4 | (count-1)
struct Bvh8 { std::vector<Wide> nodes; std::vector<int> order; };

// mode 0: the product's rule (leaf = sub-tree of <= maxLeaf triangles; open the largest-area inner child until 8 children)
// mode 1: cost-driven (Ylitie et al. 2017 dynamic programme; every wide-node visit and every leaf visit costs 1 x area)
static Bvh8 collapse(const Bvh2& b, int maxLeaf, int mode) {
    Bvh8 w; w.order = b.order; int N = (int)b.nodes.size();
    std::vector<int> cnt(N, 0);
    {   // iterative post-order for counts
        std::vector<int> st{b.root}, po; while (!st.empty()) { int id = st.back(); st.pop_back(); po.push_back(id); if (!b.nodes[id].count) { st.push_back(b.nodes[id].l); st.push_back(b.nodes[id].r); } }
        for (int k = (int)po.size() - 1; k >= 0; k--) { int id = po[k]; cnt[id] = b.nodes[id].count ? b.nodes[id].count : cnt[b.nodes[id].l] + cnt[b.nodes[id].r]; }
        if (mode == 1) {
            // C[id][i], i = 1..7: min cost of representing sub-tree id with at most i roots; D[j], j = 2..8: split j roots over the two children
            std::vector<std::array<float, 8>> C(N); std::vector<std::array<signed char, 9>> dk(N); std::vector<std::array<signed char, 8>> kind(N);      // kind: 0 single root, 1 distribute, 2 as with one root fewer
            for (int k = (int)po.size() - 1; k >= 0; k--) {
                int id = po[k]; const Node& nd = b.nodes[id]; float A = nd.box.area();
                if (cnt[id] <= maxLeaf) { for (int i = 1; i <= 7; i++) { C[id][i] = A; kind[id][i] = 0; } continue; }
                float D[9];
                for (int j = 2; j <= 8; j++) { D[j] = 3e38f; for (int a = 1; a < j; a++) { if (a > 7 || j - a > 7) continue; float c = C[nd.l][a] + C[nd.r][j - a]; if (c < D[j]) { D[j] = c; dk[id][j] = (signed char)a; } } }
                C[id][1] = A + D[8]; kind[id][1] = 0;
                for (int i = 2; i <= 7; i++) { C[id][i] = C[id][i - 1]; kind[id][i] = 2; if (D[i] < C[id][i]) { C[id][i] = D[i]; kind[id][i] = 1; } }
            }
            struct Job { int wide, id; }; w.nodes.emplace_back(); std::vector<Job> jobs{{0, b.root}};
            while (!jobs.empty()) {
                Job j = jobs.back(); jobs.pop_back();
                std::vector<int> roots;
                std::function<void(int, int)> place = [&](int id, int i) {          // sub-tree id with budget i
                    while (kind[id][i] == 2) i--;
                    if (kind[id][i] == 0) { roots.push_back(id); return; }
                    int a = dk[id][i]; place(b.nodes[id].l, a); place(b.nodes[id].r, i - a);
                };
                { int a = dk[j.id][8]; place(b.nodes[j.id].l, a); place(b.nodes[j.id].r, 8 - a); }
                Wide wn; wn.n = 0;
                for (int id : roots) {
                    if (wn.n >= 8) { fprintf(stderr, "collapse: too many roots\n"); exit(1); }
                    wn.cb[wn.n] = b.nodes[id].box;
                    if (cnt[id] <= maxLeaf) wn.ref[wn.n] = ~((b.nodes[id].first << 4) | (cnt[id] - 1));
                    else { int nw = (int)w.nodes.size(); w.nodes.emplace_back(); wn.ref[wn.n] = nw; jobs.push_back({nw, id}); }
                    wn.n++;
                }
                w.nodes[j.wide] = wn;
            }
            return w;
        }
    }
    struct Job { int wide, id; }; w.nodes.emplace_back(); std::vector<Job> jobs{{0, b.root}};
    while (!jobs.empty()) {
        Job j = jobs.back(); jobs.pop_back();
        int ids[8]; int n = 0; ids[n++] = b.nodes[j.id].l; ids[n++] = b.nodes[j.id].r;
        while (n < 8) { int best = -1; float ba = -1; for (int k = 0; k < n; k++) if (cnt[ids[k]] > maxLeaf) { float a = b.nodes[ids[k]].box.area(); if (a > ba) { ba = a; best = k; } }
            if (best < 0) break; int id = ids[best]; ids[best] = b.nodes[id].l; ids[n++] = b.nodes[id].r; }
        Wide wn; wn.n = n;
        for (int k = 0; k < n; k++) { int id = ids[k]; wn.cb[k] = b.nodes[id].box;
            if (cnt[id] <= maxLeaf) wn.ref[k] = ~((b.nodes[id].first << 4) | (cnt[id] - 1));
            else { int nw = (int)w.nodes.size(); w.nodes.emplace_back(); wn.ref[k] = nw; jobs.push_back({nw, id}); } }
        w.nodes[j.wide] = wn;
    }
    return w;
}

struct Ray { V3 o, d; };
struct Hit { float t; int prim; };
struct Ctr { uint64_t nodes = 0, leaves = 0, tris = 0, rays = 0; };

static inline bool isect(const Tri& tr, V3 o, V3 d, float tmax, float& t) {
    V3 p = cross(d, tr.e2); float det = dot(tr.e1, p); if (det == 0.f) return false; float inv = 1.f / det;
    V3 tv = o - tr.v0; float u = dot(tv, p) * inv; if (u < 0.f || u > 1.f) return false;
    V3 q = cross(tv, tr.e1); float v = dot(d, q) * inv; if (v < 0.f || u + v > 1.f) return false;
    t = dot(tr.e2, q) * inv; return t > 0.f && t < tmax;
}
static Hit trace8(const Bvh8& w, const Ray& r, Ctr& c) {
    Hit h{1e30f, -1}; V3 id{1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z};
    struct E { int ref; float t; }; E stack[256]; int sp = 0; int cur = 0; bool have = true;
    c.rays++;
    while (true) {
        if (!have) { bool got = false; while (sp) { E e = stack[--sp]; if (e.t <= h.t) { cur = e.ref; got = true; break; } } if (!got) break; }
        have = false;
        if (cur >= 0) {
            const Wide& n = w.nodes[cur]; c.nodes++;
            E hits[8]; int nh = 0;
            for (int k = 0; k < n.n; k++) {
                const Box& b = n.cb[k];
                float tx1 = (b.mn.x - r.o.x) * id.x, tx2 = (b.mx.x - r.o.x) * id.x, ty1 = (b.mn.y - r.o.y) * id.y, ty2 = (b.mx.y - r.o.y) * id.y, tz1 = (b.mn.z - r.o.z) * id.z, tz2 = (b.mx.z - r.o.z) * id.z;
                float tn = std::max(std::max(std::min(tx1, tx2), std::min(ty1, ty2)), std::max(std::min(tz1, tz2), 0.f));
                float tf = std::min(std::min(std::max(tx1, tx2), std::max(ty1, ty2)), std::min(std::max(tz1, tz2), h.t));
                if (tn <= tf) hits[nh++] = {n.ref[k], tn};
            }
            if (!nh) continue;
            std::sort(hits, hits + nh, [](const E& a, const E& b) { return a.t < b.t; });
            for (int k = nh - 1; k >= 1; k--) stack[sp++] = hits[k];
            cur = hits[0].ref; have = true;
        } else {
            int code = ~cur, first = code >> 4, cnt = (code & 15) + 1; c.leaves++; c.tris += cnt;
            for (int k = 0; k < cnt; k++) { int t = w.order[first + k]; float tt; if (isect(tris[t], r.o, r.d, h.t, tt)) { h.t = tt; h.prim = t; } }
        }
    }
    return h;
}

static uint32_t rng_state = 12345u;
static inline float frand(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (s >> 8) * (1.0f / 16777216.0f); }

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/bvh_lab_tris.bin";
    int nPaths = argc > 2 ? atoi(argv[2]) : 150000;
    FILE* f = fopen(path, "rb"); if (!f) { perror(path); return 1; }
    uint32_t n; if (fread(&n, 4, 1, f) != 1) return 1; std::vector<float> raw((size_t)n * 9); if (fread(raw.data(), 36, n, f) != n) return 1; fclose(f);
    tris.resize(n); tbox.resize(n); tcen.resize(n);
    for (uint32_t i = 0; i < n; i++) { V3 a{raw[9 * i], raw[9 * i + 1], raw[9 * i + 2]}, b{raw[9 * i + 3], raw[9 * i + 4], raw[9 * i + 5]}, c{raw[9 * i + 6], raw[9 * i + 7], raw[9 * i + 8]};
        tris[i] = {a, b - a, c - a}; Box bx; bx.grow(a); bx.grow(b); bx.grow(c); tbox[i] = bx; tcen[i] = a + ((b - a) + (c - a)) * (1.0f / 3.0f); }
    fprintf(stderr, "%u triangles\n", n);

    // rays: camera rays + cosine bounces over the baseline tree
    Bvh2 lb = build_lbvh(); Bvh8 base = collapse(lb, 4, 0);
    std::vector<Ray> rays;
    { V3 pos{4.0f, 1.7f, 20.0f}, dir = normalize(V3{1.0f, 0.12f, 0.05f}), up{0, 1, 0}; V3 U = normalize(cross(dir, up)), Vv = normalize(cross(U, dir)); float th = std::tan(0.5f * 1.0471975512f), asp = 16.f / 9.f;
      uint32_t s = 777u; Ctr dummy;
      for (int p = 0; p < nPaths; p++) {
          float sx = frand(s) * 2 - 1, sy = frand(s) * 2 - 1;
          Ray r{pos, normalize(dir + U * (sx * th * asp) + Vv * (sy * th))};
          for (int bounce = 0; bounce < 9; bounce++) {
              rays.push_back(r);
              Hit h = trace8(base, r, dummy); if (h.prim < 0) break;
              if (bounce >= 2 && frand(s) > 0.75f) break;
              const Tri& tr = tris[h.prim]; V3 ng = normalize(cross(tr.e1, tr.e2)); if (dot(ng, r.d) > 0) ng = ng * -1.f;
              V3 hp = r.o + r.d * h.t + ng * 1e-4f;
              float u1 = frand(s), u2 = frand(s), rr = std::sqrt(u1), ph = 6.2831853f * u2; V3 t1 = normalize(std::fabs(ng.x) > 0.5f ? cross(ng, V3{0, 1, 0}) : cross(ng, V3{1, 0, 0})), t2 = cross(ng, t1);
              r = {hp, normalize(t1 * (rr * std::cos(ph)) + t2 * (rr * std::sin(ph)) + ng * std::sqrt(std::max(0.f, 1 - u1)))};
          }
      } }
    fprintf(stderr, "%zu rays\n", rays.size());
    auto eval = [&](const char* name, const Bvh2& b2, int maxLeaf, int mode) {
        Bvh8 w = collapse(b2, maxLeaf, mode);
        Ctr tot; uint64_t N = 0, L = 0, T = 0;
        #pragma omp parallel
        { Ctr c;
          #pragma omp for schedule(dynamic, 1024)
          for (size_t i = 0; i < rays.size(); i++) trace8(w, rays[i], c);
          #pragma omp critical
          { N += c.nodes; L += c.leaves; T += c.tris; } }
        size_t nl = 0, nch = 0; for (auto& wn : w.nodes) { nch += wn.n; for (int k = 0; k < wn.n; k++) nl += wn.ref[k] < 0; }
        printf("%-34s sah2 %8.2f | wide nodes %8zu fill %.2f leaves %8zu tris/leaf %.2f | per ray: nodes %6.2f leaves %6.2f tris %6.2f  (nodes+leaves %6.2f)\n", name, sah_cost2(b2), w.nodes.size(), (double)nch / w.nodes.size(), nl, (double)n / nl,
               (double)N / rays.size(), (double)L / rays.size(), (double)T / rays.size(), (double)(N + L) / rays.size());
        fflush(stdout);
    };
    if (!getenv("LAB_SKIP_BASE")) {
    eval("lbvh greedy leaf4 (product)", lb, 4, 0);
    eval("lbvh cost-driven leaf4", lb, 4, 1);
    eval("lbvh greedy leaf8", lb, 8, 0);
    for (int r : {16, 64}) { Bvh2 p = build_ploc(r); char nm[64]; snprintf(nm, 64, "ploc r=%d greedy leaf4", r); eval(nm, p, 4, 0); snprintf(nm, 64, "ploc r=%d cost-driven leaf4", r); eval(nm, p, 4, 1); }
    }
    if (getenv("LAB_SAH")) { Bvh2 s = build_sah(32); eval("binned sah greedy leaf4", s, 4, 0); eval("binned sah cost-driven leaf4", s, 4, 1); eval("binned sah cost-driven leaf8", s, 8, 1); }
    if (getenv("LAB_PRODUCT")) { Bvh2 s = build_product(); eval("product builder cost-driven leaf4", s, 4, 1); }
    if (getenv("LAB_REINS")) { Bvh2 s = build_sah(32); eval("binned sah cost-driven leaf4", s, 4, 1);
        for (int it = 0; it < 3; it++) { optimize_reinsert(s, 2, 0.25f); char nm[64]; snprintf(nm, 64, "sah + reinsertion x%d cost leaf4", 2 * (it + 1)); eval(nm, s, 4, 1); } }
    if (getenv("LAB_PARALLEL_REINS")) { Bvh2 s = build_ploc(32); eval("ploc r=32 cost-driven leaf4", s, 4, 1); const float frac = getenv("LAB_FRACTION") ? (float)atof(getenv("LAB_FRACTION")) : 1.0f;
        for (int it = 0; it < (getenv("LAB_ROUNDS") ? atoi(getenv("LAB_ROUNDS")) : 4); it++) { optimize_parallel_reinsert(s, 2, frac); char nm[64]; snprintf(nm, 64, "ploc + parallel reinsertion x%d", 2 * (it + 1)); eval(nm, s, 4, 1); } }
    if (getenv("LAB_SBVH_REINS")) { Bvh2 s = build_sbvh(32, 1e-5f, 1.5f); eval("sbvh cost-driven leaf4", s, 4, 1); optimize_reinsert(s, 2, 0.25f); eval("sbvh + reinsertion x2 cost leaf4", s, 4, 1); }
    if (getenv("LAB_SBVH")) for (float al : {1e-5f, 1e-6f}) { Bvh2 s = build_sbvh(32, al, 1.5f); char nm[64]; snprintf(nm, 64, "sbvh a=%g cost-driven leaf4", al); eval(nm, s, 4, 1); }
    return 0;
}
