// mi355pt — the "prefer fast trace" tree (the reference builds its acceleration structures with AccelStructBuildFlags::PreferFastTrace, Rtxpt/Sample.cpp:1093):
// a binned surface-area-heuristic top-down build over the world-space triangles, run on the host's cores while the scene is being set up. It only decides the
// TOPOLOGY (leaf order + binary hierarchy); bounds, leaves, the BVH8 collapse, quantisation, alpha records and every refit stay on the GPU (pt_build.hip), exactly
// as for the PLOC tree, and the closest hit does not depend on the tree (pt_scene.h tri_box_accepts). tools/bvh_lab: 14-18 % fewer node visits per ray than PLOC
// on C3. Animated rebuilds (pt_animate rebuild = 1) use PLOC: 15 ms instead of a few hundred.
//
// Layout contract with pt_build.hip (the same the Karras / PLOC builders fulfil): triangles in depth-first leaf order (order[]), inner node 0 is the root,
// childL / childR hold an inner node id or BVH_LEAF_BIT | leaf position, every inner node covers the contiguous leaf range [rangeFirst, rangeLast].
// Inner node ids need no allocator: a node whose left sub-tree ends at leaf position m - 1 is node m - 1 (one id per gap between two neighbouring leaves), with the
// ids of the root and of gap 0 swapped.
#include "pt_build_sah.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ptk {
namespace {

const uint kLeafBit = 0x80000000u;
const int kBins = 32;
struct B3 { float mn[3], mx[3];
    void reset() { mn[0] = mn[1] = mn[2] = FLT_MAX; mx[0] = mx[1] = mx[2] = -FLT_MAX; }
    void grow(const B3& o) { for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); } }
    void growP(const float* p) { for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
    float area() const { float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2]; return (x < 0.f) ? 0.f : x * y + y * z + z * x; } };
struct Bins { B3 box[3][kBins]; uint cnt[3][kBins]; void reset() { for (int a = 0; a < 3; a++) for (int i = 0; i < kBins; i++) { box[a][i].reset(); cnt[a][i] = 0; } } };

// persistent workers: run(f) calls f(t) on every worker t and returns when all are done. The top levels issue a few hundred of these with little work each, so the
// workers SPIN on a generation counter while the pool lives (the length of one build): waking 100 sleeping threads through a condition variable cost more
// than the work (3.6 s instead of 0.1 s for 2.8 M triangles on a 128-core host).
struct Pool {
    unsigned T; std::vector<std::thread> th; const std::function<void(unsigned)>* job = nullptr;
    std::atomic<unsigned long long> gen{0}; std::atomic<unsigned> pending{0}; std::atomic<bool> quit{false};
    std::mutex errLock; std::exception_ptr err;                 // first exception thrown by a job on any thread (allocation failures happen inside the workers): run() rethrows it on the caller's thread
    void guarded(unsigned t) { try { (*job)(t); } catch (...) { std::lock_guard<std::mutex> g(errLock); if (!err) err = std::current_exception(); } }
    explicit Pool(unsigned n) : T(n) {
        try { for (unsigned t = 1; t < T; t++) th.emplace_back([this, t] { unsigned long long seen = 0;
            for (;;) { unsigned spins = 0;
                       while (gen.load(std::memory_order_acquire) == seen) { if (quit.load(std::memory_order_relaxed)) return; if (++spins > 2000u) std::this_thread::yield(); }
                       seen++; guarded(t); pending.fetch_sub(1, std::memory_order_acq_rel); } }); }
        catch (...) { quit.store(true); for (auto& x : th) x.join(); throw; }      // a thread could not be created: release the ones that were (the caller falls back)
    }
    ~Pool() { quit.store(true); for (auto& x : th) x.join(); }
    void run(const std::function<void(unsigned)>& f) {
        if (T == 1) { f(0); return; }
        job = &f; pending.store(T - 1, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_release);
        guarded(0);
        unsigned spins = 0; while (pending.load(std::memory_order_acquire) != 0u) if (++spins > 2000u) std::this_thread::yield();
        if (err) { std::exception_ptr e = err; err = nullptr; std::rethrow_exception(e); }      // every worker has finished the job: the caller (bvh_sah) falls back to the device builder
    }
};

struct Builder {
    const SahTri* tri; uint n; SahTopology out; uint maxLeaf; uint optimisePasses = 3u;
    std::vector<uint> scratch;                                      // partition buffer of the parallel top levels
    unsigned threads; Pool* pool;
    static uint node_id(uint gap) { return gap; }                   // while building, node = gap; relabel_root swaps the root into id 0 at the end

    struct Split { int axis; int bin; float lo, k; };
    // best (axis, bin) of the range by binned SAH; false: no axis separates the centroids (all equal)
    static bool choose(const Bins& b, const B3& cb, uint cnt, Split& s) {
        float best = FLT_MAX; s.axis = -1;
        for (int a = 0; a < 3; a++) {
            if (!(cb.mx[a] > cb.mn[a])) continue;
            float right[kBins]; B3 acc; acc.reset(); uint c = 0;
            for (int i = kBins - 1; i > 0; i--) { acc.grow(b.box[a][i]); c += b.cnt[a][i]; right[i] = c ? acc.area() * (float)c : FLT_MAX; }
            acc.reset(); c = 0;
            for (int i = 0; i < kBins - 1; i++) {
                acc.grow(b.box[a][i]); c += b.cnt[a][i];
                if (!c || c == cnt) continue;
                float cost = acc.area() * (float)c + right[i + 1];
                if (cost < best) { best = cost; s.axis = a; s.bin = i; s.lo = cb.mn[a]; s.k = (float)kBins / (cb.mx[a] - cb.mn[a]); }
            }
        }
        return s.axis >= 0;
    }
    static int bin_of(float c, float lo, float k) { int b = (int)((c - lo) * k); return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b); }

    void emit(uint lo, uint mid, uint hi, uint parentId) {          // inner node over [lo, hi) split at mid
        const uint id = node_id(mid - 1u);
        out.rangeFirst[id] = lo; out.rangeLast[id] = hi - 1u; out.parent[id] = parentId;
    }

    // sequential build of [lo, hi) (its inner node's parent is parentId; returns the split position)
    uint build_serial(uint lo, uint hi, uint parentId) {
        struct Job { uint lo, hi, parent; bool right; };
        std::vector<Job> stack; stack.push_back({lo, hi, parentId, false});
        uint topMid = 0; bool first = true;
        uint* order = out.order;
        while (!stack.empty()) {
            Job j = stack.back(); stack.pop_back();
            const uint cnt = j.hi - j.lo;
            uint mid;
            // position splits work on ranges sorted by triangle id: the tree — which the optimiser below refines down to single triangles — must not depend on the
            // order a partition happens to leave behind (it differs between the parallel and the serial partition, i.e. between thread counts)
            if (cnt <= maxLeaf) { std::sort(order + j.lo, order + j.hi); mid = j.lo + cnt / 2u; }
            else {
                B3 cb; cb.reset();
                for (uint i = j.lo; i < j.hi; i++) cb.growP(tri[order[i]].c);
                Bins b; b.reset();
                float k3[3]; for (int a = 0; a < 3; a++) k3[a] = (cb.mx[a] > cb.mn[a]) ? (float)kBins / (cb.mx[a] - cb.mn[a]) : 0.f;
                for (uint i = j.lo; i < j.hi; i++) { const SahTri& t = tri[order[i]]; B3 tb; memcpy(tb.mn, t.mn, 12); memcpy(tb.mx, t.mx, 12);
                    for (int a = 0; a < 3; a++) { int bi = bin_of(t.c[a], cb.mn[a], k3[a]); b.box[a][bi].grow(tb); b.cnt[a][bi]++; } }
                Split s;
                if (!choose(b, cb, cnt, s)) { std::sort(order + j.lo, order + j.hi); mid = j.lo + cnt / 2u; }
                else {
                    mid = (uint)(std::partition(order + j.lo, order + j.hi, [&](uint t) { return bin_of(tri[t].c[s.axis], s.lo, s.k) <= s.bin; }) - order);
                    if (mid == j.lo || mid == j.hi) { std::sort(order + j.lo, order + j.hi); mid = j.lo + cnt / 2u; }
                }
            }
            emit(j.lo, mid, j.hi, j.parent);
            const uint id = node_id(mid - 1u);
            if (first) { topMid = mid; first = false; }
            else (j.right ? out.childR : out.childL)[j.parent] = id;
            // children: leaves are linked here, inner children link themselves when they are popped
            if (mid - j.lo == 1u) { out.childL[id] = kLeafBit | j.lo; out.leafParent[j.lo] = id; } else stack.push_back({j.lo, mid, id, false});
            if (j.hi - mid == 1u) { out.childR[id] = kLeafBit | mid; out.leafParent[mid] = id; } else stack.push_back({mid, j.hi, id, true});
        }
        return topMid;
    }

    template <class F> void parallel_for(uint lo, uint hi, F f) {   // f(thread, begin, end)
        const unsigned T = threads; const uint len = hi - lo;
        pool->run([&](unsigned t) { uint a = lo + (uint)((unsigned long long)len * t / T), b = lo + (uint)((unsigned long long)len * (t + 1) / T); if (b > a) f(t, a, b); });
    }
    // one big node, all threads: returns the split position (order[lo, hi) is partitioned in place, stably per side)
    uint split_parallel(uint lo, uint hi) {
        const uint cnt = hi - lo; uint* order = out.order;
        std::vector<B3> cbs(threads); for (auto& c : cbs) c.reset();
        parallel_for(lo, hi, [&](unsigned t, uint a, uint b) { B3 cb; cb.reset(); for (uint i = a; i < b; i++) cb.growP(tri[order[i]].c); cbs[t] = cb; });
        B3 cb; cb.reset(); for (auto& c : cbs) cb.grow(c);
        float k3[3]; for (int a = 0; a < 3; a++) k3[a] = (cb.mx[a] > cb.mn[a]) ? (float)kBins / (cb.mx[a] - cb.mn[a]) : 0.f;
        std::vector<Bins> bs(threads); for (auto& b : bs) b.reset();
        parallel_for(lo, hi, [&](unsigned t, uint a, uint e) { Bins& b = bs[t];
            for (uint i = a; i < e; i++) { const SahTri& tr = tri[order[i]]; B3 tb; memcpy(tb.mn, tr.mn, 12); memcpy(tb.mx, tr.mx, 12);
                for (int ax = 0; ax < 3; ax++) { int bi = bin_of(tr.c[ax], cb.mn[ax], k3[ax]); b.box[ax][bi].grow(tb); b.cnt[ax][bi]++; } } });
        Bins all; all.reset();
        for (auto& b : bs) for (int a = 0; a < 3; a++) for (int i = 0; i < kBins; i++) { all.box[a][i].grow(b.box[a][i]); all.cnt[a][i] += b.cnt[a][i]; }     // (thread order: min / max / integer sums are exact)
        Split s;
        if (!choose(all, cb, cnt, s)) { std::sort(order + lo, order + hi); return lo + cnt / 2u; }
        // stable two-sided partition through the scratch buffer: per-thread counts, prefix, scatter
        std::vector<uint> nl(threads + 1, 0u);
        parallel_for(lo, hi, [&](unsigned t, uint a, uint e) { uint c = 0; for (uint i = a; i < e; i++) c += bin_of(tri[order[i]].c[s.axis], s.lo, s.k) <= s.bin ? 1u : 0u; nl[t + 1] = c; });
        for (unsigned t = 0; t < threads; t++) nl[t + 1] += nl[t];
        const uint nLeft = nl[threads];
        if (nLeft == 0u || nLeft == cnt) { std::sort(order + lo, order + hi); return lo + cnt / 2u; }
        parallel_for(lo, hi, [&](unsigned t, uint a, uint e) { uint l = lo + nl[t], r = lo + nLeft + (a - lo) - nl[t];
            for (uint i = a; i < e; i++) { uint v = order[i]; if (bin_of(tri[v].c[s.axis], s.lo, s.k) <= s.bin) scratch[l++] = v; else scratch[r++] = v; } });
        parallel_for(lo, hi, [&](unsigned, uint a, uint e) { memcpy(order + a, scratch.data() + a, 4u * (size_t)(e - a)); });
        return lo + nLeft;
    }

    void run() {
        for (uint i = 0; i < n; i++) out.order[i] = i;
        if (n == 1u) return;
        scratch.resize(n);
        // top of the tree: ranges above `grain` are split by all threads together, breadth first; what is left goes to a task list
        const uint grain = std::max(8192u, n / (threads * 2u));
        struct Range { uint lo, hi, mid; int parentSlot; bool right, parallel; };
        std::vector<Range> top; top.push_back({0u, n, 0u, -1, false, false});
        for (size_t k = 0; k < top.size(); k++) {
            Range r = top[k];
            if (r.hi - r.lo <= grain || threads == 1u) continue;
            top[k].mid = split_parallel(r.lo, r.hi); top[k].parallel = true;
            top.push_back({r.lo, top[k].mid, 0u, (int)k, false, false}); top.push_back({top[k].mid, r.hi, 0u, (int)k, true, false});
        }
        auto parent_of = [&](const Range& r) { return r.parentSlot < 0 ? 0xFFFFFFFFu : node_id(top[(size_t)r.parentSlot].mid - 1u); };
        // the serial sub-trees (largest first), then the parallel-split nodes on top of them
        std::vector<size_t> tasks; for (size_t k = 0; k < top.size(); k++) if (!top[k].parallel && top[k].hi - top[k].lo > 1u) tasks.push_back(k);
        std::sort(tasks.begin(), tasks.end(), [&](size_t a, size_t b) { uint sa = top[a].hi - top[a].lo, sb = top[b].hi - top[b].lo; return sa != sb ? sa > sb : a < b; });
        std::atomic<size_t> next(0);
        pool->run([&](unsigned) { for (;;) { size_t i = next.fetch_add(1); if (i >= tasks.size()) break; Range& r = top[tasks[i]]; r.mid = build_serial(r.lo, r.hi, parent_of(r)); } });
        for (const Range& r : top) if (r.parallel) { const uint id = node_id(r.mid - 1u); out.rangeFirst[id] = r.lo; out.rangeLast[id] = r.hi - 1u; out.parent[id] = parent_of(r); }
        for (const Range& r : top) {                                // link every range of the top part into its parent
            if (r.parentSlot < 0) continue;
            const uint pid = parent_of(r);
            if (r.hi - r.lo == 1u) { (r.right ? out.childR : out.childL)[pid] = kLeafBit | r.lo; out.leafParent[r.lo] = pid; }
            else (r.right ? out.childR : out.childL)[pid] = node_id(r.mid - 1u);
        }
        uint root = node_id(top[0].mid - 1u);
        if (optimisePasses) optimise(root, optimisePasses, 0.25f);
        relabel_root(root);
        if (out.absorb) choose_wide_nodes();
    }
    // ---- insertion-based optimisation of the finished binary tree (Bittner, Hapala & Havran 2013; batched as in Meister & Bittner 2018): a node whose parent's
    // box is large is taken out of the tree and put back where it adds the least surface area. Per pass: (1) every candidate searches the FROZEN tree in
    // parallel — branch and bound from the root over the tree with the candidate removed (its ancestors' boxes shrunk accordingly); (2) the moves are applied
    // one after the other, best gain first, skipping a move that touches a node an earlier move of the pass touched or that would now close a cycle;
    // (3) all inner boxes are recomputed. No reference is duplicated and the leaf set is unchanged, so the hit definition (pt_scene.h) is untouched.
    // tools/bvh_lab on C3: SAH cost 151 -> 121, wide-node visits per ray 15.4 -> 13.2, leaf visits 3.7 -> 2.9, triangle tests 11.9 -> 9.8.
    // Node ids while optimising: inner nodes keep their ids [0, n - 1), leaf at position q is n - 1 + q.
    struct Move { float gain; uint x, target; };
    void optimise(uint& root, uint passes, float fraction) {
        if (n < 8u) return;
        const uint N = 2u * n - 1u, I = n - 1u;
        std::vector<B3> box(N); std::vector<uint> L(I), R(I), par(N, 0xFFFFFFFFu);
        auto ref_id = [&](uint ref) { return (ref & kLeafBit) ? I + (ref & ~kLeafBit) : ref; };
        parallel_for(0u, I, [&](unsigned, uint a, uint b) { for (uint i = a; i < b; i++) { L[i] = ref_id(out.childL[i]); R[i] = ref_id(out.childR[i]); } });
        for (uint i = 0; i < I; i++) { par[L[i]] = i; par[R[i]] = i; }
        parallel_for(0u, n, [&](unsigned, uint a, uint b) { for (uint q = a; q < b; q++) { const SahTri& t = tri[out.order[q]]; memcpy(box[I + q].mn, t.mn, 12); memcpy(box[I + q].mx, t.mx, 12); } });
        std::vector<uint> post; post.reserve(I);
        auto refit_all = [&]() {                                     // pre-order list of the inner nodes, boxes in reverse
            post.clear(); std::vector<uint> st{root};
            while (!st.empty()) { uint id = st.back(); st.pop_back(); post.push_back(id); if (L[id] < I) st.push_back(L[id]); if (R[id] < I) st.push_back(R[id]); }
            for (size_t k = post.size(); k-- > 0;) { const uint id = post[k]; B3 b = box[L[id]]; b.grow(box[R[id]]); box[id] = b; }
        };
        refit_all();
        std::vector<float> score; std::vector<uint> cand; std::vector<Move> moves; std::vector<std::vector<Move>> found(threads); std::vector<std::vector<uint>> picked(threads);
        std::vector<uint> mark(N, 0u); uint epoch = 0u;
        double tSel = 0, tFilter = 0, tSearch = 0, tApply = 0; auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        auto refit_up = [&](uint a) { for (; a != 0xFFFFFFFFu; a = par[a]) { B3 nb = box[L[a]]; nb.grow(box[R[a]]); if (!memcmp(&nb, &box[a], sizeof(B3))) break; box[a] = nb; } };
        const uint kClasses = 6u;                                    // a pass visits the candidates in six hashed classes: a node, its parent, its sibling rarely share one, so few moves collide
        for (uint pass = 0; pass < passes; pass++) {
            // candidates of the pass: the nodes under the largest parents (the top `fraction` by the parent's area)
            double t0 = now();
            score.clear();
            for (uint x = 0; x < N; x++) { const uint p = par[x]; if (p != 0xFFFFFFFFu && par[p] != 0xFFFFFFFFu) score.push_back(box[p].area()); }
            size_t take = (size_t)((double)score.size() * fraction); if (take < 1u) take = 1u; if (take > score.size()) take = score.size();
            std::nth_element(score.begin(), score.begin() + (take - 1u), score.end(), [](float a, float b) { return a > b; });
            const float threshold = score[take - 1u];
            size_t proposed = 0, applied = 0; tSel += now() - t0;
            for (uint cls = 0; cls < kClasses; cls++) {
                double t1 = now();
                for (auto& v : picked) v.clear();
                parallel_for(0u, N, [&](unsigned t, uint a, uint b) { for (uint x = a; x < b; x++) { const uint p = par[x];
                    if (p == 0xFFFFFFFFu || par[p] == 0xFFFFFFFFu || ((x * 2654435761u) >> 13) % kClasses != cls || !(box[p].area() >= threshold)) continue; picked[t].push_back(x); } });
                cand.clear(); for (auto& v : picked) cand.insert(cand.end(), v.begin(), v.end());
                const size_t nc = cand.size(); double t2 = now(); tFilter += t2 - t1;
                for (auto& f : found) f.clear();
                std::atomic<size_t> next(0);
                pool->run([&](unsigned t) {
                    struct Q { float induced; uint node; int path; };
                    auto cmp = [](const Q& a, const Q& b) { return a.induced > b.induced; };
                    std::vector<Q> heap; std::vector<uint> path; std::vector<B3> pbox;
                    for (;;) {
                        const size_t c0 = next.fetch_add(256u); if (c0 >= nc) break;
                        for (size_t ci = c0; ci < std::min(nc, c0 + 256u); ci++) {
                            const uint x = cand[ci], p = par[x], g = par[p], s = (L[p] == x) ? R[p] : L[p];
                            const B3 xb = box[x]; const float xa = xb.area();
                            // the tree without x and p: s hangs under g; boxes of g .. root shrink
                            path.clear(); pbox.clear();
                            { B3 cur = box[s]; uint below = p;
                              for (uint a = g; a != 0xFFFFFFFFu; a = par[a]) { const uint other = (L[a] == below) ? R[a] : L[a]; B3 nb = box[other]; nb.grow(cur); path.push_back(a); pbox.push_back(nb); cur = nb; below = a; } }
                            float stay = 0.f;
                            for (size_t k = 0; k < path.size(); k++) { B3 u = pbox[k]; u.grow(xb); stay += u.area() - pbox[k].area(); }
                            { B3 u = box[s]; u.grow(xb); stay += u.area(); }
                            heap.clear(); heap.push_back({0.f, path.back(), (int)path.size() - 1});
                            float best = FLT_MAX; uint bestNode = s; uint steps = 0;
                            while (!heap.empty() && ++steps <= 8192u) {      // (branch and bound ends far earlier on real scenes; the cap bounds the time on adversarial ones)
                                std::pop_heap(heap.begin(), heap.end(), cmp); const Q q = heap.back(); heap.pop_back();
                                if (q.induced + xa >= best) break;
                                const B3& nb = q.path >= 0 ? pbox[(size_t)q.path] : box[q.node];
                                B3 u = nb; u.grow(xb); const float tot = q.induced + u.area();
                                if (tot < best) { best = tot; bestNode = q.node; }
                                if (q.node >= I) continue;
                                const float ind = tot - nb.area();
                                if (!(ind + xa < best)) continue;
                                for (int side = 0; side < 2; side++) {
                                    uint c = side ? R[q.node] : L[q.node]; int cp = -1;
                                    if (q.path >= 0) { const uint on = q.path > 0 ? path[(size_t)q.path - 1u] : p; if (c == on) { if (q.path == 0) c = s; else cp = q.path - 1; } }
                                    heap.push_back({ind, c, cp}); std::push_heap(heap.begin(), heap.end(), cmp);
                                }
                            }
                            if (bestNode != s && bestNode != path.back() && stay - best > 1e-6f * stay) found[t].push_back({stay - best, x, bestNode});
                        }
                    }
                });
                double t3 = now(); tSearch += t3 - t2;
                moves.clear(); for (auto& f : found) moves.insert(moves.end(), f.begin(), f.end());
                std::sort(moves.begin(), moves.end(), [](const Move& a, const Move& b) { return a.gain != b.gain ? a.gain > b.gain : a.x < b.x; });
                epoch++; proposed += moves.size();
                for (const Move& m : moves) {
                    const uint x = m.x, p = par[x], g = par[p], s = (L[p] == x) ? R[p] : L[p], t = m.target, tp = par[t];
                    if (tp == 0xFFFFFFFFu || g == 0xFFFFFFFFu) continue;
                    if (mark[x] == epoch || mark[p] == epoch || mark[s] == epoch || mark[g] == epoch || mark[t] == epoch || mark[tp] == epoch) continue;
                    bool cycle = false; for (uint a = t; a != 0xFFFFFFFFu; a = par[a]) if (a == x) { cycle = true; break; }      // (an earlier move of the pass may have put the target below x)
                    if (cycle) continue;
                    mark[x] = mark[p] = mark[s] = mark[g] = mark[t] = mark[tp] = epoch;
                    (L[g] == p ? L[g] : R[g]) = s; par[s] = g;                      // take p (and x below it) out
                    (L[tp] == t ? L[tp] : R[tp]) = p; par[p] = tp;                  // p goes where the target was, with the target and x below it
                    L[p] = t; R[p] = x; par[t] = p; par[x] = p; applied++;
                    refit_up(g); { B3 nb = box[t]; nb.grow(box[x]); box[p] = nb; } refit_up(tp);
                }
                tApply += now() - t3;
            }
            if (getenv("MI355PT_SAH_DEBUG")) { refit_all(); double c = 0; for (uint id : post) c += box[id].area(); fprintf(stderr, "  optimise pass %u: %zu candidates, %zu proposed, %zu applied, inner area / root area %.2f | select %.3f filter %.3f search %.3f apply %.3f s so far\n", pass, take, proposed, applied, c / box[root].area(), tSel, tFilter, tSearch, tApply); }
        }
        // back to the layout contract: depth-first leaf order, inner node = the gap after its left sub-tree
        std::vector<uint> newOrder; newOrder.reserve(n); std::vector<uint> nid(I), npos(n), first(I);
        { struct F { uint node; uint stage; }; std::vector<F> st; st.push_back({root, 0u});
          while (!st.empty()) { F f = st.back(); st.pop_back();
              if (f.node >= I) { npos[f.node - I] = (uint)newOrder.size(); newOrder.push_back(out.order[f.node - I]); continue; }
              if (f.stage == 0u) { first[f.node] = (uint)newOrder.size(); st.push_back({f.node, 1u}); st.push_back({L[f.node], 0u}); }
              else if (f.stage == 1u) { nid[f.node] = (uint)newOrder.size() - 1u; st.push_back({f.node, 2u}); st.push_back({R[f.node], 0u}); }
              else { const uint id = nid[f.node]; out.rangeFirst[id] = first[f.node]; out.rangeLast[id] = (uint)newOrder.size() - 1u; } } }
        auto out_ref = [&](uint node) { return node >= I ? (kLeafBit | npos[node - I]) : nid[node]; };
        parallel_for(0u, I, [&](unsigned, uint a, uint b) { for (uint i = a; i < b; i++) { const uint id = nid[i]; out.childL[id] = out_ref(L[i]); out.childR[id] = out_ref(R[i]);
                                                                                                  out.parent[id] = par[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : nid[par[i]]; } });
        parallel_for(0u, n, [&](unsigned, uint a, uint b) { for (uint q = a; q < b; q++) out.leafParent[npos[q]] = nid[par[I + q]]; });
        memcpy(out.order, newOrder.data(), 4u * (size_t)n);
        root = nid[root];
    }
    // Wide-node assignment (after relabel_root). C[id][i], i = 1..7: least cost of representing the sub-tree of inner node id by at most i roots (a root is a leaf or
    // a wide node; a wide node costs its area plus the best split of 8 roots over its two children). absorb[id] = the parent's wide node opens id.
    void choose_wide_nodes() {
        const uint N = n - 1u;
        std::vector<B3> box(N); std::vector<float> C((size_t)N * 8u); std::vector<unsigned char> dk((size_t)N * 9u), kind((size_t)N * 8u);
        auto count_of = [&](uint id) { return out.rangeLast[id] - out.rangeFirst[id] + 1u; };
        auto child_box = [&](uint ref) { if (ref & kLeafBit) { const SahTri& t = tri[out.order[ref & ~kLeafBit]]; B3 b; memcpy(b.mn, t.mn, 12); memcpy(b.mx, t.mx, 12); return b; } return box[ref]; };
        auto cost = [&](uint ref, uint i, float leafArea) { return (ref & kLeafBit) ? leafArea : C[(size_t)ref * 8u + i]; };
        auto dp_node = [&](uint id) {
            const uint L = out.childL[id], R = out.childR[id];
            B3 bl = child_box(L), br = child_box(R), b = bl; b.grow(br); box[id] = b;
            const float A = b.area(); float* c = &C[(size_t)id * 8u]; unsigned char* kd = &kind[(size_t)id * 8u];
            if (count_of(id) <= maxLeaf) { for (uint i = 1; i <= 7; i++) { c[i] = A; kd[i] = 0; } return; }
            const float al = bl.area(), ar = br.area(); float D[9];
            for (uint j = 2; j <= 8; j++) { D[j] = FLT_MAX; for (uint a = 1; a < j; a++) { if (a > 7u || j - a > 7u) continue; float v = cost(L, a, al) + cost(R, j - a, ar); if (v < D[j]) { D[j] = v; dk[(size_t)id * 9u + j] = (unsigned char)a; } } }
            c[1] = A + D[8]; kd[1] = 0;
            for (uint i = 2; i <= 7; i++) { c[i] = c[i - 1]; kd[i] = 2; if (D[i] < c[i]) { c[i] = D[i]; kd[i] = 1; } }
        };
        auto mark_node = [&](uint id, uint budget, bool wideRoot, std::vector<uint>& st) {      // st: (id, budget << 1 | wideRoot) pairs still to visit
            uint i = budget;
            if (!wideRoot) {
                if (count_of(id) <= maxLeaf) return;                                            // a leaf of the wide tree
                while (kind[(size_t)id * 8u + i] == 2) i--;
                if (kind[(size_t)id * 8u + i] == 0) { st.push_back(id); st.push_back((8u << 1) | 1u); return; }      // a root of the forest: a wide node of its own
                out.absorb[id] = 1u;
            }
            const uint a = dk[(size_t)id * 9u + i], L = out.childL[id], R = out.childR[id];
            if (!(L & kLeafBit)) { st.push_back(L); st.push_back(a << 1); }
            if (!(R & kLeafBit)) { st.push_back(R); st.push_back((i - a) << 1); }
        };
        // the nodes above `grain` triangles are done by one thread (parents are discovered before their children: reverse order = bottom up), the sub-trees below in parallel
        const uint grain = std::max(8192u, n / (threads * 8u));
        std::vector<uint> topNodes, taskRoots;
        { std::vector<uint> st{0u}; while (!st.empty()) { uint id = st.back(); st.pop_back();
            if (count_of(id) <= grain) { taskRoots.push_back(id); continue; }
            topNodes.push_back(id);
            if (!(out.childL[id] & kLeafBit)) st.push_back(out.childL[id]); if (!(out.childR[id] & kLeafBit)) st.push_back(out.childR[id]); } }
        std::atomic<size_t> next(0);
        pool->run([&](unsigned) { std::vector<uint> po, st; for (;;) { size_t t = next.fetch_add(1); if (t >= taskRoots.size()) break;
            po.clear(); st.assign(1, taskRoots[t]);
            while (!st.empty()) { uint id = st.back(); st.pop_back(); po.push_back(id); if (!(out.childL[id] & kLeafBit)) st.push_back(out.childL[id]); if (!(out.childR[id] & kLeafBit)) st.push_back(out.childR[id]); }
            for (size_t k = po.size(); k-- > 0;) dp_node(po[k]); } });
        for (size_t k = topNodes.size(); k-- > 0;) dp_node(topNodes[k]);
        // top down: which nodes are opened inside their parent's wide node. The top part serially, collecting the entry state of every task root; then the tasks in parallel.
        pool->run([&](unsigned t) { const uint a = (uint)((unsigned long long)N * t / threads), b = (uint)((unsigned long long)N * (t + 1u) / threads); for (uint i = a; i < b; i++) out.absorb[i] = 0u; });
        std::vector<uint> entry(N, 0xFFFFFFFFu);                     // task root -> budget << 1 | wideRoot when the marking reaches it
        { std::vector<uint> st{0u, (8u << 1) | 1u};
          while (!st.empty()) { uint state = st.back(); st.pop_back(); uint id = st.back(); st.pop_back();
              if (count_of(id) <= grain) { entry[id] = state; continue; }                        // a task root (the collection above stopped at exactly these nodes)
              mark_node(id, state >> 1, state & 1u, st); } }
        next.store(0);
        pool->run([&](unsigned) { std::vector<uint> st; for (;;) { size_t t = next.fetch_add(1); if (t >= taskRoots.size()) break;
            const uint root = taskRoots[t]; if (entry[root] == 0xFFFFFFFFu) continue;
            st.assign({root, entry[root]});
            while (!st.empty()) { uint state = st.back(); st.pop_back(); uint id = st.back(); st.pop_back(); mark_node(id, state >> 1, state & 1u, st); } } });
    }
    // the tree was built with identity ids (node = gap): swap the labels of `gap` and 0 so that the root becomes node 0
    void relabel_root(uint gap) {
        if (gap == 0u) return;
        auto sw = [&](uint* a) { std::swap(a[0], a[gap]); };
        sw(out.childL); sw(out.childR); sw(out.rangeFirst); sw(out.rangeLast); sw(out.parent);
        auto fix = [&](uint& r) { if (r == 0xFFFFFFFFu || (r & kLeafBit)) return; if (r == gap) r = 0u; else if (r == 0u) r = gap; };
        for (uint i = 0; i + 1u < n; i++) { fix(out.childL[i]); fix(out.childR[i]); fix(out.parent[i]); }
        for (uint i = 0; i < n; i++) { uint& p = out.leafParent[i]; if (p == gap) p = 0u; else if (p == 0u) p = gap; }
    }
};

} // namespace

uint bvh_sah_topology(const SahTri* tris, uint n, const SahTopology& out, uint maxLeaf, unsigned threads) {
    if (n == 0u) return 0u;
    Builder b; b.tri = tris; b.n = n; b.out = out; b.maxLeaf = maxLeaf ? maxLeaf : 1u;
    unsigned hw = std::thread::hardware_concurrency(); if (!hw) hw = 8;
    b.threads = threads ? threads : std::min(hw, 32u);               // the build is bound by gathers from the triangle array; beyond a few dozen threads the sync costs more than it buys (and 8 ranks of a node build at once)
    if (b.threads > n / 4096u + 1u) b.threads = n / 4096u + 1u;
    if (const char* e = getenv("MI355PT_SAH_OPTIMISE")) b.optimisePasses = (uint)atoi(e);      // developer A/B: 0 = the plain binned-SAH tree
    Pool pool(b.threads); b.pool = &pool;
    b.run();
    return n >= 8u ? b.optimisePasses : 0u;
}

} // namespace ptk
