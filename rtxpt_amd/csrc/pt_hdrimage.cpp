// mi355pt — float image files for the environment source. The reference takes ".exr", ".hdr" and ".dds" environment maps (Rtxpt/Sample.cpp:116) through
// Donut's TextureCache (EnvMapBaker.cpp:392-415), which is not vendored in the reference tree: the two HDR formats are read here from their published
// specifications. Host code, no device.
//   OpenEXR: single-part scan-line files, channels R G B (or Y) of type half or float, compression NONE / RLE / ZIPS / ZIP. Tiled, multi-part and deep
//            files and the PIZ / PXR24 / B44 / DWA codecs are reported as PT_ERROR_UNSUPPORTED.
//   Radiance .hdr: "#?RADIANCE" / "#?RGBE", FORMAT=32-bit_rle_rgbe, -Y h +X w; flat and new-style run-length scan lines.
// Output: width x height x 3 floats, top row first (the first scan line of either format is the top of the picture).
#include "../../include/mi355pt.h"
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

bool read_file(const char* path, std::vector<unsigned char>& out) {
    FILE* f = fopen(path, "rb"); if (!f) return false;
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return false; }
    long n = ftell(f); if (n < 0 || n > (1l << 31)) { fclose(f); return false; }
    rewind(f); out.resize((size_t)n);
    bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n; fclose(f); return ok;
}
float half_to_float(unsigned h) {
    const unsigned s = (h >> 15) & 1u, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu; unsigned bits;
    if (e == 0) { if (m == 0) bits = s << 31; else { int k = 0; unsigned mm = m; while (!(mm & 0x400u)) { mm <<= 1; k++; } bits = (s << 31) | ((unsigned)(113 - k) << 23) | ((mm & 0x3FFu) << 13); } }
    else if (e == 31) bits = (s << 31) | 0x7F800000u | (m << 13);
    else bits = (s << 31) | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
struct Rd { const unsigned char* p; size_t n, i; bool ok;
    unsigned char u8() { if (i + 1 > n) { ok = false; return 0; } return p[i++]; }
    int i32() { if (i + 4 > n) { ok = false; return 0; } int v; memcpy(&v, p + i, 4); i += 4; return v; }
    unsigned long long u64() { if (i + 8 > n) { ok = false; return 0; } unsigned long long v; memcpy(&v, p + i, 8); i += 8; return v; }
    bool cstr(std::string& s, size_t maxLen) { s.clear(); while (i < n && p[i]) { if (s.size() >= maxLen) { ok = false; return false; } s.push_back((char)p[i++]); } if (i >= n) { ok = false; return false; } i++; return true; } };

// OpenEXR's byte predictor and de-interleave, undone (ImfZip.cpp / ImfRle.cpp: the codec stores the first half of the bytes then the second half, as deltas)
void exr_unpredict_interleave(std::vector<unsigned char>& t, std::vector<unsigned char>& out) {
    const size_t n = t.size(); out.resize(n);
    for (size_t k = 1; k < n; k++) t[k] = (unsigned char)(t[k - 1] + t[k] - 128);
    const size_t half = (n + 1) / 2;
    for (size_t k = 0; k < n; k++) out[k] = (k & 1u) ? t[half + k / 2] : t[k / 2];
}
bool exr_rle_decode(const unsigned char* in, size_t nin, std::vector<unsigned char>& out, size_t want) {
    out.clear(); out.reserve(want); size_t i = 0;
    while (i < nin) {
        const int c = (signed char)in[i++];
        if (c < 0) { size_t cnt = (size_t)(-c); if (i + cnt > nin || out.size() + cnt > want) return false; out.insert(out.end(), in + i, in + i + cnt); i += cnt; }
        else { size_t cnt = (size_t)c + 1; if (i >= nin || out.size() + cnt > want) return false; out.insert(out.end(), cnt, in[i++]); }
    }
    return out.size() == want;
}

int32_t read_exr(const std::vector<unsigned char>& d, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    Rd r{d.data(), d.size(), 0, true};
    if (d.size() < 8 || (unsigned)r.i32() != 20000630u) return PT_ERROR_IO;
    const unsigned ver = (unsigned)r.i32();
    if ((ver & 0xFFu) != 2u) return PT_ERROR_UNSUPPORTED;
    if (ver & (0x200u | 0x800u | 0x1000u)) return PT_ERROR_UNSUPPORTED;           // tiled, deep, multi-part
    const size_t maxName = (ver & 0x400u) ? 255 : 31;
    struct Chan { std::string name; int type, xs, ys; };
    std::vector<Chan> ch; int comp = -1, dw[4] = {0, 0, -1, -1}, lineOrder = 0; bool haveDw = false;
    for (;;) {
        std::string name, type; if (!r.cstr(name, maxName)) return PT_ERROR_IO;
        if (name.empty()) break;
        if (!r.cstr(type, maxName)) return PT_ERROR_IO;
        const int sz = r.i32(); if (!r.ok || sz < 0 || r.i + (size_t)sz > r.n) return PT_ERROR_IO;
        Rd a{r.p + r.i, (size_t)sz, 0, true}; r.i += (size_t)sz;
        if (name == "channels" && type == "chlist") {
            for (;;) { Chan c; if (!a.cstr(c.name, maxName)) return PT_ERROR_IO; if (c.name.empty()) break;
                c.type = a.i32(); a.u8(); a.u8(); a.u8(); a.u8(); c.xs = a.i32(); c.ys = a.i32(); if (!a.ok) return PT_ERROR_IO; ch.push_back(c); if (ch.size() > 64) return PT_ERROR_UNSUPPORTED; }
        } else if (name == "compression") comp = a.u8();
        else if (name == "dataWindow" && type == "box2i") { for (int k = 0; k < 4; k++) dw[k] = a.i32(); haveDw = a.ok; }
        else if (name == "lineOrder") lineOrder = a.u8();
    }
    if (ch.empty() || comp < 0 || !haveDw) return PT_ERROR_IO;
    if (comp > 3) return PT_ERROR_UNSUPPORTED;                                      // 0 none, 1 RLE, 2 ZIPS, 3 ZIP; PIZ / PXR24 / B44 / DWA are not read
    const long long w = (long long)dw[2] - dw[0] + 1, h = (long long)dw[3] - dw[1] + 1;
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768 || w * h > (1ll << 28)) return PT_ERROR_IO;      // (a damaged data window must not turn into a 12 GB allocation)
    size_t lineBytes = 0; int idx[3] = {-1, -1, -1}, yIdx = -1; std::vector<size_t> chOff(ch.size());
    for (size_t k = 0; k < ch.size(); k++) {
        if (ch[k].xs != 1 || ch[k].ys != 1) return PT_ERROR_UNSUPPORTED;            // sub-sampled (luminance / chroma) channels
        if (ch[k].type < 0 || ch[k].type > 2) return PT_ERROR_IO;
        chOff[k] = lineBytes; lineBytes += (size_t)w * (ch[k].type == 1 ? 2u : 4u);
        if (ch[k].name == "R") idx[0] = (int)k; else if (ch[k].name == "G") idx[1] = (int)k; else if (ch[k].name == "B") idx[2] = (int)k; else if (ch[k].name == "Y") yIdx = (int)k;
    }
    if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0) { if (yIdx < 0) return PT_ERROR_UNSUPPORTED; idx[0] = idx[1] = idx[2] = yIdx; }
    for (int k = 0; k < 3; k++) if (ch[(size_t)idx[k]].type == 0) return PT_ERROR_UNSUPPORTED;      // uint channels carry ids, not radiance
    const unsigned linesPerBlock = comp == 3 ? 16u : 1u;
    const size_t blocks = ((size_t)h + linesPerBlock - 1) / linesPerBlock;
    if (blocks * 8 > d.size()) return PT_ERROR_IO;                                   // the offset table alone would not fit the file
    std::vector<unsigned long long> offs(blocks); for (auto& o : offs) o = r.u64();
    if (!r.ok) return PT_ERROR_IO;
    W = (uint32_t)w; H = (uint32_t)h; rgb.assign((size_t)w * h * 3, 0.f);
    std::vector<unsigned char> raw, tmp;
    (void)lineOrder;                                                                // every block carries its own y: the order of the blocks in the file does not matter
    for (size_t b = 0; b < blocks; b++) {
        if (offs[b] > d.size() || d.size() - (size_t)offs[b] < 8) return PT_ERROR_IO;
        Rd c{d.data(), d.size(), (size_t)offs[b], true};
        const long long y0 = (long long)c.i32() - dw[1]; const int size = c.i32();
        if (!c.ok || size < 0 || c.i + (size_t)size > c.n || y0 < 0 || y0 >= h) return PT_ERROR_IO;
        const size_t lines = (size_t)std::min<long long>(linesPerBlock, h - y0), want = lines * lineBytes;
        const unsigned char* src = d.data() + c.i;
        if ((size_t)size == want) raw.assign(src, src + want);                      // stored as is (also what the codecs fall back to when they do not shrink the block)
        else if (comp == 0) return PT_ERROR_IO;
        else if (comp == 1) { if (!exr_rle_decode(src, (size_t)size, tmp, want)) return PT_ERROR_IO; exr_unpredict_interleave(tmp, raw); }
        else { tmp.resize(want); uLongf got = (uLongf)want; if (uncompress(tmp.data(), &got, src, (uLong)size) != Z_OK || got != want) return PT_ERROR_IO; exr_unpredict_interleave(tmp, raw); }
        for (size_t l = 0; l < lines; l++) {
            const unsigned char* line = raw.data() + l * lineBytes; float* o = &rgb[((size_t)(y0 + (long long)l) * (size_t)w) * 3];
            for (int k = 0; k < 3; k++) {
                const Chan& cc = ch[(size_t)idx[k]]; const unsigned char* p = line + chOff[(size_t)idx[k]];
                if (cc.type == 1) for (size_t x = 0; x < (size_t)w; x++) { unsigned short v; memcpy(&v, p + 2 * x, 2); o[3 * x + (size_t)k] = half_to_float(v); }
                else for (size_t x = 0; x < (size_t)w; x++) { float v; memcpy(&v, p + 4 * x, 4); o[3 * x + (size_t)k] = v; }
            }
        }
    }
    return PT_OK;
}

int32_t read_rgbe(const std::vector<unsigned char>& d, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    size_t i = 0; auto line = [&](std::string& s) { s.clear(); while (i < d.size() && d[i] != '\n') { if (s.size() > 4096) return false; s.push_back((char)d[i++]); } if (i >= d.size()) return false; i++; return true; };
    std::string s; if (!line(s) || (s.compare(0, 10, "#?RADIANCE") != 0 && s.compare(0, 6, "#?RGBE") != 0)) return PT_ERROR_IO;
    bool fmt = false;
    for (;;) { if (!line(s)) return PT_ERROR_IO; if (s.empty()) break; if (s.compare(0, 7, "FORMAT=") == 0) { if (s != "FORMAT=32-bit_rle_rgbe") return PT_ERROR_UNSUPPORTED; fmt = true; } }
    (void)fmt;
    if (!line(s)) return PT_ERROR_IO;
    // resolution string: "-Y h +X w" is the standard orientation (top row first, left to right); the flipped variants are read too, the transposed ones ("+X w -Y h" ...) are not
    long h = 0, w = 0; char sy = 0, sx = 0;
    if (sscanf(s.c_str(), "%cY %ld %cX %ld", &sy, &h, &sx, &w) != 4 || (sy != '-' && sy != '+') || (sx != '-' && sx != '+')) return PT_ERROR_UNSUPPORTED;
    const bool flipY = sy == '+', flipX = sx == '-';
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768 || (long long)w * h > (1ll << 28)) return PT_ERROR_IO;
    if ((size_t)h > d.size()) return PT_ERROR_IO;                                    // every scan line takes at least a byte
    W = (uint32_t)w; H = (uint32_t)h; rgb.assign((size_t)w * h * 3, 0.f);
    std::vector<unsigned char> sl((size_t)w * 4);
    for (long y = 0; y < h; y++) {
        if (i + 4 > d.size()) return PT_ERROR_IO;
        if (w >= 8 && w < 32768 && d[i] == 2 && d[i + 1] == 2 && (((unsigned)d[i + 2] << 8) | d[i + 3]) == (unsigned)w) {      // new-style run-length: four planes
            i += 4;
            for (int c = 0; c < 4; c++) { long x = 0;
                while (x < w) { if (i >= d.size()) return PT_ERROR_IO; unsigned cnt = d[i++];
                    if (cnt > 128) { cnt -= 128; if (cnt == 0 || x + cnt > (unsigned long)w || i >= d.size()) return PT_ERROR_IO; unsigned char v = d[i++]; for (unsigned k = 0; k < cnt; k++) sl[(size_t)(x++) * 4 + (size_t)c] = v; }
                    else { if (cnt == 0 || x + cnt > (unsigned long)w || i + cnt > d.size()) return PT_ERROR_IO; for (unsigned k = 0; k < cnt; k++) sl[(size_t)(x++) * 4 + (size_t)c] = d[i++]; } } }
        } else { if (i + (size_t)w * 4 > d.size()) return PT_ERROR_IO; memcpy(sl.data(), d.data() + i, (size_t)w * 4); i += (size_t)w * 4; }
        float* o = &rgb[(size_t)(flipY ? h - 1 - y : y) * (size_t)w * 3];
        for (long x = 0; x < w; x++) { const unsigned char* p = &sl[(size_t)(flipX ? w - 1 - x : x) * 4];
            if (p[3] == 0) { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = 0.f; }
            else { const float f = ldexpf(1.0f, (int)p[3] - (128 + 8)); o[3 * x] = (float)p[0] * f; o[3 * x + 1] = (float)p[1] * f; o[3 * x + 2] = (float)p[2] * f; } }      // (stb_image / Radiance: mantissa x 2^(e - 136), no +0.5)
    }
    return PT_OK;
}

} // namespace

extern "C" int32_t pt_image_read_float(const char* path, uint32_t* width, uint32_t* height, float** rgbOut) {
    if (!path || !width || !height || !rgbOut) return PT_ERROR_INVALID_ARGUMENT;
    *rgbOut = nullptr; *width = *height = 0;
    try {
        std::vector<unsigned char> d; if (!read_file(path, d)) return PT_ERROR_IO;
        std::vector<float> rgb; uint32_t w = 0, h = 0; int32_t r;
        if (d.size() >= 4 && d[0] == 0x76 && d[1] == 0x2f && d[2] == 0x31 && d[3] == 0x01) r = read_exr(d, w, h, rgb);
        else if (d.size() >= 2 && d[0] == '#' && d[1] == '?') r = read_rgbe(d, w, h, rgb);
        else if (d.size() >= 4 && !memcmp(d.data(), "DDS ", 4)) {          // float .dds (RGBA16F / RGBA32F, what the reference's "save baked cube" and nvtt write for HDR images): pt_dds.cpp
            uint32_t fmt = 0; void* px = nullptr; r = pt_image_read_dds(path, &w, &h, &fmt, &px);
            if (r == PT_OK && fmt != PT_TEX_RGBA32F) { free(px); return PT_ERROR_UNSUPPORTED; }      // an 8-bit image is no environment source
            if (r == PT_OK) { rgb.resize((size_t)w * h * 3u); const float* q = (const float*)px; for (size_t i = 0; i < (size_t)w * h; i++) { rgb[3 * i] = q[4 * i]; rgb[3 * i + 1] = q[4 * i + 1]; rgb[3 * i + 2] = q[4 * i + 2]; } free(px); }
        }
        else return PT_ERROR_UNSUPPORTED;
        if (r != PT_OK) return r;
        float* out = (float*)malloc(rgb.size() * sizeof(float)); if (!out) return PT_ERROR_HIP;
        memcpy(out, rgb.data(), rgb.size() * sizeof(float));
        *rgbOut = out; *width = w; *height = h;
        return PT_OK;
    } catch (...) { return PT_ERROR_IO; }
}
extern "C" void pt_image_free(float* rgb) { free(rgb); }
