#include "/root/repo/include/mi355pt.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static unsigned long long s = 88172645463325252ull; static unsigned rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(int argc, char** argv) {
    long ok = 0, bad = 0; int iters = atoi(argv[1]);
    for (int a = 2; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb"); std::vector<unsigned char> seed; int c; while ((c = fgetc(f)) != EOF) seed.push_back((unsigned char)c); fclose(f);
        bool isJpg = strstr(argv[a], ".jpg") != nullptr;
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> d = seed; int nm = 1 + rnd() % 6;
            for (int m = 0; m < nm; m++) { unsigned k = rnd() % 4, i = rnd() % d.size();
                if (k == 0) d[i] = (unsigned char)rnd(); else if (k == 1) d[i] ^= 1u << (rnd() % 8); else if (k == 2) { d[i] = 0xFF; if (i + 1 < d.size()) d[i + 1] = (unsigned char)(0xC0 + rnd() % 0x30); } else if (d.size() > 8) d.resize(4 + rnd() % (d.size() - 4)); }
            unsigned w = 0, h = 0, fmt = 0; void* px = nullptr; int r;
            if (isJpg) r = pt_image_read_jpeg(d.data(), d.size(), &w, &h, &px); else r = pt_image_read_dds_memory(d.data(), d.size(), &w, &h, &fmt, &px);
            if (r == 0) { ok++; volatile unsigned char t = ((unsigned char*)px)[(size_t)w * h * (fmt == 2 ? 16 : 4) - 1]; (void)t; pt_image_free((float*)px); } else bad++;
        }
    }
    printf("decoded %ld, refused %ld\n", ok, bad); return 0;
}
