#include "/root/repo/include/mi355pt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" { int32_t pt_set_materials(pt_context*, const PTMaterialData*, uint32_t, const PtTextureDesc*, uint32_t) { return 0; }
int32_t pt_set_geometry(pt_context*, const PtGeometryBuffers*, const PtGeometryDesc*, uint32_t, const PtMeshDesc*, uint32_t) { return 0; }
int32_t pt_set_instances(pt_context*, const PtInstanceDesc*, uint32_t) { return 0; }
int32_t pt_set_lights(pt_context*, const PolymorphicLightInfo*, const PolymorphicLightInfoEx*, uint32_t) { return 0; } }
static unsigned long long s = 0x1234567ull; static unsigned rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(int argc, char** argv) {
    int iters = atoi(argv[1]); long ok = 0, bad = 0;
    for (int a = 2; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb"); std::vector<unsigned char> seed; int c; while ((c = fgetc(f)) != EOF) seed.push_back((unsigned char)c); fclose(f);
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> d = seed; int nm = 1 + rnd() % 4;
            for (int m = 0; m < nm; m++) { unsigned k = rnd() % 3, i = rnd() % d.size(); if (k == 0) d[i] = (unsigned char)rnd(); else if (k == 1) d[i] ^= 1u << (rnd() % 8); else if (d.size() > 40) d.resize(33 + rnd() % (d.size() - 33)); }
            FILE* o = fopen("g/x.png", "wb"); fwrite(d.data(), 1, d.size(), o); fclose(o);
            pt_scene_import* S = nullptr; PtSceneJsonInfo info; int r = pt_scene_json_import("g/t.scene.json", nullptr, &S, &info);
            if (r == 0) { if (info.numTextures) { ok++; PtTextureDesc t; pt_scene_import_texture(S, 0, &t); volatile unsigned char v = ((const unsigned char*)t.pixels)[(size_t)t.width * t.height * 4 - 1]; (void)v; } else bad++; pt_scene_import_free(S); } else bad++;
        }
    }
    printf("decoded %ld, refused %ld\n", ok, bad); return 0;
}
