// test harness (CPU only) for rtxpt_amd/csrc/pt_build_sah.cpp: builds the "prefer fast trace" topology over generated triangle soups and checks the layout
// contract pt_build.hip relies on. usage: bvh_sah_check <n> <mode> <threads> <seed>; mode 0 uniform soup, 1 all centroids equal, 2 long thin slivers on a line,
// 3 clustered duplicates. Prints "ok <wide nodes> <children>" or a diagnosis and exits 1.
#include "../rtxpt_amd/csrc/pt_build_sah.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace ptk;
static unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static float frand(unsigned& s) { return (float)rnd(s) / 16777216.0f; }
int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const uint n = (uint)atoi(argv[1]); const int mode = atoi(argv[2]); const unsigned threads = (unsigned)atoi(argv[3]); unsigned seed = (unsigned)atoi(argv[4]);
    std::vector<SahTri> t(n);
    for (uint i = 0; i < n; i++) {
        float c[3], h[3];
        for (int a = 0; a < 3; a++) { c[a] = frand(seed) * 100.f; h[a] = frand(seed) * 0.5f; }
        if (mode == 1) { c[0] = c[1] = c[2] = 7.f; }
        if (mode == 2) { c[1] = c[2] = 0.f; h[0] = 60.f; h[1] = h[2] = 1e-4f; }
        if (mode == 3) { for (int a = 0; a < 3; a++) c[a] = (float)(rnd(seed) % 3u); }
        for (int a = 0; a < 3; a++) { t[i].mn[a] = c[a] - h[a]; t[i].mx[a] = c[a] + h[a]; t[i].c[a] = c[a]; }
    }
    const uint m = n ? n : 1u;
    std::vector<uint> order(m, 0xFFFFFFFFu), cl(m), cr(m), rf(m), rl(m), par(m), lp(m), ab(m, 7u);
    bvh_sah_topology(t.data(), n, SahTopology{order.data(), cl.data(), cr.data(), rf.data(), rl.data(), par.data(), lp.data(), ab.data()}, 4u, threads);
    if (n == 0) { printf("ok 0 0\n"); return 0; }
    std::vector<char> seen(n, 0);
    for (uint i = 0; i < n; i++) { if (order[i] >= n || seen[order[i]]) { printf("order is not a permutation at %u\n", i); return 1; } seen[order[i]] = 1; }
    if (n == 1) { printf("ok 0 0\n"); return 0; }
    size_t inner = 0, leaves = 0; std::vector<uint> st{0u};
    while (!st.empty()) {
        uint id = st.back(); st.pop_back(); inner++;
        if (id >= n - 1) { printf("inner id %u out of range\n", id); return 1; }
        const uint L = cl[id], R = cr[id];
        const uint lf = (L >> 31) ? (L & 0x7FFFFFFFu) : rf[L], ll = (L >> 31) ? (L & 0x7FFFFFFFu) : rl[L], rf2 = (R >> 31) ? (R & 0x7FFFFFFFu) : rf[R], rl2 = (R >> 31) ? (R & 0x7FFFFFFFu) : rl[R];
        if (lf != rf[id] || ll + 1 != rf2 || rl2 != rl[id]) { printf("node %u covers [%u,%u] but its children cover [%u,%u] [%u,%u]\n", id, rf[id], rl[id], lf, ll, rf2, rl2); return 1; }
        for (uint c : {L, R}) {
            if (c >> 31) { leaves++; if (lp[c & 0x7FFFFFFFu] != id) { printf("leafParent mismatch\n"); return 1; } }
            else { if (par[c] != id) { printf("parent mismatch\n"); return 1; } st.push_back(c); }
        }
        if (ab[id] > 1u) { printf("absorb flag of node %u not written\n", id); return 1; }
    }
    if (inner != n - 1 || leaves != n || par[0] != 0xFFFFFFFFu) { printf("inner %zu leaves %zu root parent %u\n", inner, leaves, par[0]); return 1; }
    // k_collapse8 in cost-driven mode: open the absorbed inner children; a wide node may never end up with more than 8
    auto isLeaf = [&](uint ref) { return (ref >> 31) || (rl[ref] - rf[ref] + 1 <= 4); };
    size_t wide = 0, kids = 0; std::vector<uint> roots{0u};
    if (ab[0]) { printf("the root is marked absorbed\n"); return 1; }
    while (!roots.empty()) {
        uint r = roots.back(); roots.pop_back(); wide++;
        std::vector<uint> fr{cl[r], cr[r]}; bool again = true;
        while (again) { again = false; for (size_t k = 0; k < fr.size(); k++) if (!isLeaf(fr[k]) && ab[fr[k]]) { uint id = fr[k]; fr[k] = cl[id]; fr.push_back(cr[id]); again = true; break; } }
        if (fr.size() > 8) { printf("wide node with %zu children\n", fr.size()); return 1; }
        kids += fr.size();
        for (uint c : fr) if (!isLeaf(c)) roots.push_back(c);
    }
    unsigned long long h = 1469598103934665603ull;                      // FNV-1a over the whole topology: equal trees for equal inputs, whatever the thread count
    auto mix = [&](const std::vector<uint>& v, size_t k) { for (size_t i = 0; i < k; i++) { h ^= v[i]; h *= 1099511628211ull; } };
    mix(order, n); mix(cl, n - 1); mix(cr, n - 1); mix(rf, n - 1); mix(rl, n - 1); mix(par, n - 1); mix(lp, n); mix(ab, n - 1);
    // surface-area cost of the binary tree over the generated boxes (inner nodes, relative to the root): what the optimiser lowers
    std::vector<float> bmn((size_t)(n - 1) * 3), bmx((size_t)(n - 1) * 3); std::vector<uint> po; { std::vector<uint> s2{0u}; while (!s2.empty()) { uint id = s2.back(); s2.pop_back(); po.push_back(id); if (!(cl[id] >> 31)) s2.push_back(cl[id]); if (!(cr[id] >> 31)) s2.push_back(cr[id]); } }
    double cost = 0, rootArea = 0;
    for (size_t k = po.size(); k-- > 0;) { uint id = po[k]; float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
        for (uint c : {cl[id], cr[id]}) for (int a = 0; a < 3; a++) { float lo = (c >> 31) ? t[order[c & 0x7FFFFFFFu]].mn[a] : bmn[(size_t)c * 3 + a], hi = (c >> 31) ? t[order[c & 0x7FFFFFFFu]].mx[a] : bmx[(size_t)c * 3 + a]; mn[a] = std::fmin(mn[a], lo); mx[a] = std::fmax(mx[a], hi); }
        for (int a = 0; a < 3; a++) { bmn[(size_t)id * 3 + a] = mn[a]; bmx[(size_t)id * 3 + a] = mx[a]; }
        double ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2], ar = ex * ey + ey * ez + ez * ex; cost += ar; if (id == 0) rootArea = ar; }
    printf("ok %zu %zu %016llx %.4f\n", wide, kids, h, rootArea > 0 ? cost / rootArea : 0.0);
    return 0;
}
