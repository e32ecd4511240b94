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
    for (int m = 0; m < nm; m++) { unsigned k = rnd() % 5, i = rnd() % d.size();
        if (k == 0) d[i] = (unsigned char)rnd(); else if (k == 1) d[i] ^= 1u << (rnd() % 8); else if (k == 2) { const char* t = toks[rnd() % 9]; d.insert(d.begin() + i, t, t + strlen(t)); } else if (k == 3 && d.size() > 16) d.resize(8 + rnd() % (d.size() - 8)); else if (d.size() > 8) d.erase(d.begin() + i, d.begin() + i + 1 + rnd() % 3 % (d.size() - i)); }
    return d; }
int main(int argc, char** argv) {
    int iters = atoi(argv[1]); long ok = 0, bad = 0;
    auto gltf = rd("g/c.gltf"), bin = rd("g/c.bin"), agltf = rd("g/a.gltf"), abin = rd("g/a.bin"), scene = rd("g/t.scene.json"), mat = rd("g/Materials/red.material.json");
    for (int it = 0; it < iters; it++) {
        int which = rnd() % 5;
        wr("g/c.gltf", which == 0 ? mutate(gltf) : gltf); wr("g/c.bin", which == 1 ? mutate(bin) : bin); wr("g/t.scene.json", which == 2 ? mutate(scene) : scene);
        wr("g/Materials/red.material.json", which == 3 ? mutate(mat) : mat); wr("g/a.gltf", which == 4 ? mutate(agltf) : agltf);
        pt_scene_import* S = nullptr; PtSceneJsonInfo info; int r = pt_scene_json_import("g/t.scene.json", nullptr, &S, &info);
        if (r == 0) { ok++; std::vector<PtInstanceDesc> I(info.numInstances + 1); pt_scene_import_instances(S, I.data(), info.numInstances); std::vector<PTMaterialData> M(info.numMaterials + 1); pt_scene_import_materials(S, M.data(), info.numMaterials);
                      for (uint32_t t = 0; t < info.numTextures; t++) { PtTextureDesc d; pt_scene_import_texture(S, t, &d); } pt_scene_import_free(S); } else bad++;
        pt_gltf_animation* A = nullptr; uint32_t na = 0; float dur = 0; r = pt_gltf_animation_load("g/a.gltf", &A, &na, &dur);
        if (r == 0) { std::vector<PtInstanceDesc> I(64); for (uint32_t a = 0; a < na && a < 3; a++) pt_gltf_animation_instances(A, a, (float)(rnd() % 300) * 0.01f - 0.5f, I.data(), 64); pt_gltf_animation_free(A); }
        { auto mv = which == 3 ? mutate(mat) : mat; std::string m(mv.begin(), mv.end()); PTMaterialData out; PtMaterialJsonInfo mi; uint32_t words[5] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; pt_material_from_json(m.c_str(), words, &out, &mi); }
    }
    printf("imported %ld, refused %ld\n", ok, bad); return 0;
}
