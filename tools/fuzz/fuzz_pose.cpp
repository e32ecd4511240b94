#include "/root/repo/include/mi355pt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
// the device half of the API is not linked here: stubs
extern "C" { int32_t pt_set_materials(pt_context*, const PTMaterialData*, uint32_t, const PtTextureDesc*, uint32_t) { return 0; }
int32_t pt_set_geometry(pt_context*, const PtGeometryBuffers*, const PtGeometryDesc*, uint32_t, const PtMeshDesc*, uint32_t) { return 0; }
int32_t pt_set_instances(pt_context*, const PtInstanceDesc*, uint32_t) { return 0; }
int32_t pt_set_lights(pt_context*, const PolymorphicLightInfo*, const PolymorphicLightInfoEx*, uint32_t) { return 0; } }
static unsigned long long s = 0x2545F4914F6CDD1Dull; static unsigned rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
static std::vector<unsigned char> rd(const char* p) { std::vector<unsigned char> v; FILE* f = fopen(p, "rb"); int c; while ((c = fgetc(f)) != EOF) v.push_back((unsigned char)c); fclose(f); return v; }
static void wr(const char* p, const std::vector<unsigned char>& v) { FILE* f = fopen(p, "wb"); fwrite(v.data(), 1, v.size(), f); fclose(f); }
static std::vector<unsigned char> mutate(std::vector<unsigned char> d) { int nm = 1 + rnd() % 4; const char* toks[] = {"-1", "99999999", "null", "[]", "{}", "\"x\"", "1e308", "true", "4294967296"};
    for (int m = 0; m < nm; m++) { unsigned k = rnd() % 4, i = rnd() % d.size();
        if (k == 0) d[i] = (unsigned char)rnd(); else if (k == 1) d[i] ^= 1u << (rnd() % 8); else if (k == 2) { const char* t = toks[rnd() % 9]; d.insert(d.begin() + i, t, t + strlen(t)); } else d.resize(i + 1); }
    return d; }
int main(int argc, char** argv) {
    int iters = atoi(argv[1]); long ok = 0, bad = 0;
    std::vector<std::vector<unsigned char>> seeds; for (int i = 2; i < argc; i++) seeds.push_back(rd(argv[i]));
    for (int it = 0; it < iters; it++) {
        wr("m.gltf", mutate(seeds[rnd() % seeds.size()]));
        pt_gltf_animation* A = nullptr; uint32_t na = 0; float dur = 0; int r = pt_gltf_animation_load("m.gltf", &A, &na, &dur);
        if (r == 0) { ok++; const float t = (float)(rnd() % 300) * 0.01f - 0.5f;
            int n = pt_gltf_animation_normals(A, 0, t, nullptr, nullptr, 0);
            if (n > 0 && n < (1 << 22)) { std::vector<uint32_t> N(n), T(n); std::vector<float> P(3 * (size_t)n); pt_gltf_animation_normals(A, 0, t, N.data(), T.data(), (uint32_t)n); pt_gltf_animation_normals(A, 0, t, N.data(), nullptr, (uint32_t)n);
                                          pt_gltf_animation_positions(A, 0, t, P.data(), (uint32_t)n); }
            pt_gltf_animation_free(A); } else bad++;
    }
    printf("posed %ld, refused %ld\n", ok, bad); return 0;
}
