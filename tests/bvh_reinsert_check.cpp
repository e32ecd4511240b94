// test harness (CPU only) for rtxpt_amd/csrc/pt_build_reinsert.h — the per-node functions of the DEVICE-side insertion-based optimiser (k_ri_search / k_ri_lock /
// k_ri_check / k_ri_ring / k_ri_apply run exactly these), executed pass by pass the way pt_build.hip schedules them: over a deliberately poor tree of a generated
// soup (median splits in input order) every pass must leave ONE tree with every node in it exactly once and consistent parent links, and the surface-area cost must
// fall. usage: bvh_reinsert_check <n> <mode> <passes> <seed>; prints "ok <cost before> <cost after> <moves>".
#include "../rtxpt_amd/csrc/pt_build_reinsert.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace ptk;
static unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static float frand(unsigned& s) { return (float)rnd(s) / 16777216.0f; }
int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const uint n = (uint)atoi(argv[1]); const int mode = atoi(argv[2]); const int passes = atoi(argv[3]); unsigned seed = (unsigned)atoi(argv[4]);
    if (n < 2) { printf("ok 0 0 0\n"); return 0; }
    const uint N = 2u * n - 1u;
    std::vector<uint> par(N, RI_NONE), left(N, RI_NONE), right(N, RI_NONE); std::vector<float> box(8 * (size_t)N, 0.f);
    RiTree t{par.data(), left.data(), right.data(), box.data(), N};
    for (uint i = 0; i < n; i++) { RiBox b; for (int a = 0; a < 3; a++) { float c = frand(seed) * 100.f, h = frand(seed) * 0.5f; if (mode == 1) c = (float)(rnd(seed) % 3u); if (mode == 2 && a == 0) h = 40.f; b.mn[a] = c - h; b.mx[a] = c + h; } ri_store(t, i, b); }
    // leaves 0 .. n - 1, inner nodes n .. 2n - 2 with the root last (PLOC's numbering): a balanced tree over the leaves in input order
    uint next = n; struct J { uint lo, hi, parent; bool rightChild; }; std::vector<J> st{{0u, n, RI_NONE, false}}; std::vector<uint> idOf;
    { // build recursively but number parents after children: collect post-order
      struct F { uint lo, hi; int stage; uint l, r; }; std::vector<F> s{{0u, n, 0, 0u, 0u}}; std::vector<uint> ret;
      while (!s.empty()) { F& f = s.back();
          if (f.hi - f.lo == 1u) { ret.push_back(f.lo); s.pop_back(); continue; }
          const uint mid = f.lo + (f.hi - f.lo) / 2u;
          if (f.stage == 0) { f.stage = 1; s.push_back({f.lo, mid, 0, 0u, 0u}); }
          else if (f.stage == 1) { f.l = ret.back(); ret.pop_back(); f.stage = 2; s.push_back({mid, f.hi, 0, 0u, 0u}); }
          else { f.r = ret.back(); ret.pop_back(); const uint id = next++; left[id] = f.l; right[id] = f.r; par[f.l] = id; par[f.r] = id; ri_refit_node(t, id); ret.push_back(id); s.pop_back(); } } }
    if (next != N) { printf("tree construction broken\n"); return 1; }
    const uint root = N - 1u;
    auto cost = [&]() { double c = 0; for (uint i = n; i < N; i++) c += ri_area(ri_load(t, i)); return c / ri_area(ri_load(t, root)); };
    auto validate = [&](std::vector<uint>& bfs) { bfs.assign(1, root); std::vector<char> seen(N, 0); seen[root] = 1;
        for (size_t k = 0; k < bfs.size(); k++) { const uint id = bfs[k]; if ((left[id] == RI_NONE) != (right[id] == RI_NONE)) return false; if (left[id] == RI_NONE) { if (id >= n) return false; continue; } if (id < n) return false;
            for (uint c : {left[id], right[id]}) { if (c >= N || seen[c] || par[c] != id) return false; seen[c] = 1; bfs.push_back(c); } }
        return bfs.size() == N && par[root] == RI_NONE; };
    std::vector<uint> bfs; if (!validate(bfs)) { printf("initial tree invalid\n"); return 1; }
    const double before = cost(); size_t moves = 0; double prev = before;
    std::vector<float> gain(N); std::vector<uint> target(N), pivot(N), ok(N); std::vector<unsigned long long> lock(N), moving(N);
    for (int pass = 0; pass < passes; pass++) {
        for (uint x = 0; x < N; x++) { gain[x] = ri_search(t, x, 0.f, target[x], pivot[x]); if (target[x] != RI_NONE && !(gain[x] > 1e-6f * ri_area(ri_load(t, par[x])))) target[x] = RI_NONE; lock[x] = moving[x] = 0ull; ok[x] = 0u; }
        for (uint x = 0; x < N; x++) if (target[x] != RI_NONE) { const unsigned long long key = ri_key(gain[x], x); ri_for_links(t, x, target[x], [&](uint a) { if (lock[a] < key) lock[a] = key; return true; }); }
        for (uint x = 0; x < N; x++) if (target[x] != RI_NONE) { const unsigned long long key = ri_key(gain[x], x); if (ri_for_links(t, x, target[x], [&](uint a) { return lock[a] == key; })) { ok[x] = 1u; moving[x] = key; } }
        for (uint x = 0; x < N; x++) if (ok[x] == 1u && ri_gives_way(t, x, target[x], pivot[x], moving.data(), ri_key(gain[x], x))) ok[x] = 2u;
        for (uint x = N; x-- > 0u;) if (ok[x] == 1u) { ri_apply(t, x, target[x]); moves++; }      // (any order: the moves commute)
        if (!validate(bfs)) { printf("pass %d left an invalid tree\n", pass); return 1; }
        for (size_t k = bfs.size(); k-- > 0;) if (left[bfs[k]] != RI_NONE) ri_refit_node(t, bfs[k]);
        const double c = cost(); if (c > prev * 1.0005) { printf("pass %d raised the cost %.4f -> %.4f\n", pass, prev, c); return 1; } prev = c;
    }
    printf("ok %.4f %.4f %zu\n", before, prev, moves);
    return 0;
}
