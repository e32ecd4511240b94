#include "/root/repo/include/mi355pt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static unsigned long long s = 0x9E3779B97F4A7C15ull; static unsigned rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(int argc, char** argv) {
    long ok = 0, bad = 0; int iters = atoi(argv[1]);
    for (int a = 2; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb"); std::vector<unsigned char> seed; int c; while ((c = fgetc(f)) != EOF) seed.push_back((unsigned char)c); fclose(f);
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> d = seed; int nm = 1 + rnd() % 5;
            for (int m = 0; m < nm; m++) { unsigned k = rnd() % 4, i = rnd() % d.size();
                if (k == 0) d[i] = (unsigned char)rnd(); else if (k == 1) d[i] ^= 1u << (rnd() % 8); else if (k == 2) { unsigned v = rnd(); memcpy(&d[i > 4 ? i - 4 : 0], &v, 4); } else if (d.size() > 16) d.resize(8 + rnd() % (d.size() - 8)); }
            FILE* o = fopen("cur.bin", "wb"); fwrite(d.data(), 1, d.size(), o); fclose(o);
            unsigned w = 0, h = 0; float* px = nullptr;
            int r = pt_image_read_float("cur.bin", &w, &h, &px);
            if (r == 0) { ok++; volatile float t = px[(size_t)w * h * 3 - 1]; (void)t; pt_image_free(px); } else bad++;
        }
    }
    printf("decoded %ld, refused %ld\n", ok, bad); return 0;
}
