// test harness (CPU only) for rtxpt_amd/csrc/pt_build_wide.h — the per-node functions of the DEVICE-side wide-node programme (k_wide_dp / k_wide_mark run exactly
// these): over the topology the host builder makes of a generated soup, the level-by-level programme must mark exactly the inner nodes the host's own
// choose_wide_nodes marks (pt_build_sah.cpp; both restate Ylitie, Karras & Laine 2017). usage: bvh_wide_check <n> <mode> <seed>; prints "ok <levels> <absorbed>".
#include "../rtxpt_amd/csrc/pt_build_sah.h"
#include "../rtxpt_amd/csrc/pt_build_wide.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace ptk;
static unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static float frand(unsigned& s) { return (float)rnd(s) / 16777216.0f; }
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const uint n = (uint)atoi(argv[1]); const int mode = atoi(argv[2]); unsigned seed = (unsigned)atoi(argv[3]);
    std::vector<SahTri> t(n);
    for (uint i = 0; i < n; i++) {
        float c[3], h[3];
        for (int a = 0; a < 3; a++) { c[a] = frand(seed) * 100.f; h[a] = frand(seed) * 0.5f; }
        if (mode == 2) { c[1] = c[2] = 0.f; h[0] = 60.f; h[1] = h[2] = 1e-4f; }
        if (mode == 3) { for (int a = 0; a < 3; a++) c[a] = (float)(rnd(seed) % 3u); }
        for (int a = 0; a < 3; a++) { t[i].mn[a] = c[a] - h[a]; t[i].mx[a] = c[a] + h[a]; t[i].c[a] = c[a]; }
    }
    if (n < 2) { printf("ok 0 0\n"); return 0; }
    std::vector<uint> order(n), cl(n), cr(n), rf(n), rl(n), par(n), lp(n), ab(n, 0u);
    bvh_sah_topology(t.data(), n, SahTopology{order.data(), cl.data(), cr.data(), rf.data(), rl.data(), par.data(), lp.data(), ab.data()}, 4u, 2u);
    // what k_node_boxes hands the programme: per inner node the boxes of its two children (unions over their leaf ranges)
    const uint I = n - 1u; std::vector<float> lmn(3 * (size_t)I), lmx(3 * (size_t)I), rmn(3 * (size_t)I), rmx(3 * (size_t)I);
    auto range_box = [&](uint a, uint b, float* mn, float* mx) { for (int k = 0; k < 3; k++) { mn[k] = 3e38f; mx[k] = -3e38f; } for (uint q = a; q <= b; q++) for (int k = 0; k < 3; k++) { const SahTri& s = t[order[q]]; if (s.mn[k] < mn[k]) mn[k] = s.mn[k]; if (s.mx[k] > mx[k]) mx[k] = s.mx[k]; } };
    for (uint id = 0; id < I; id++) { const uint L = cl[id]; const uint gamma = (L >> 31) ? (L & 0x7FFFFFFFu) : rl[L]; range_box(rf[id], gamma, &lmn[3 * (size_t)id], &lmx[3 * (size_t)id]); range_box(gamma + 1u, rl[id], &rmn[3 * (size_t)id], &rmx[3 * (size_t)id]); }
    // breadth-first levels as k_wide_levels makes them
    std::vector<uint> bfs{0u}; std::vector<size_t> start{0u};
    for (size_t d = 0; start[d] < bfs.size(); d++) { const size_t end = bfs.size();
        for (size_t k = start[d]; k < end; k++) { const uint id = bfs[k]; if (rl[id] - rf[id] + 1u <= 4u) continue; if (!(cl[id] >> 31)) bfs.push_back(cl[id]); if (!(cr[id] >> 31)) bfs.push_back(cr[id]); }
        start.push_back(end); }
    start.pop_back();                                   // (the last entry marks the end of the last level)
    std::vector<float> C(8 * (size_t)I, -1.f); std::vector<unsigned long long> dec(I, 0ull); std::vector<uint> mine(I, 0u), state(I, 0u);
    for (size_t d = start.size(); d-- > 0;) { const size_t e = d + 1 < start.size() ? start[d + 1] : bfs.size();
        for (size_t k = start[d]; k < e; k++) { const uint id = bfs[k]; wide_dp_node(id, cl[id], cr[id], rl[id] - rf[id] + 1u, 4u, &lmn[3 * (size_t)id], &lmx[3 * (size_t)id], &rmn[3 * (size_t)id], &rmx[3 * (size_t)id], C.data(), dec.data()); } }
    state[0] = wide_state(8u, true);
    for (size_t d = 0; d < start.size(); d++) { const size_t e = d + 1 < start.size() ? start[d + 1] : bfs.size();
        for (size_t k = start[d]; k < e; k++) { const uint id = bfs[k]; wide_mark_node(id, state[id], cl[id], cr[id], rl[id] - rf[id] + 1u, 4u, dec.data(), mine.data(), state.data()); } }
    size_t absorbed = 0;
    for (uint id = 0; id < I; id++) { if (mine[id] != ab[id]) { printf("node %u (count %u): device programme %u, host %u\n", id, rl[id] - rf[id] + 1u, mine[id], ab[id]); return 1; } absorbed += mine[id]; }
    printf("ok %zu %zu\n", start.size(), absorbed);
    return 0;
}
