// mi355pt — float image files for the environment source. The reference takes ".exr", ".hdr" and ".dds" environment maps (Rtxpt/Sample.cpp:116) through
// Donut's TextureCache (EnvMapBaker.cpp:392-415), which is not vendored in the reference tree: the two HDR formats are read here from their published
// specifications. Host code, no device.
//   OpenEXR: single-part scan-line files, channels R G B (or Y) of type half or float, compression NONE / RLE / ZIPS / ZIP / PIZ (the default of most HDRI
//            tools). Single-part TILED files: level (0, 0) of any level mode, the same codecs per tile. Multi-part and deep files and the PXR24 / B44 / DWA codecs are reported as PT_ERROR_UNSUPPORTED.
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


// ---- OpenEXR's PIZ codec, decode side (ImfPizCompressor.cpp, ImfHuf.cpp, ImfWav.cpp of the OpenEXR library, restated from the published format): per block of up to 32 scan
// lines — a bitmap of the 16-bit values that occur (-> a reverse lookup table), then a Huffman-coded stream (canonical codes, lengths packed in 6 bits with zero runs, one
// run-length symbol) of the block's 16-bit words, channel by channel, each channel plane wavelet-transformed (two-dimensional Haar with 14- or 16-bit modular arithmetic).
const int HUF_ENCBITS = 16, HUF_DECBITS = 14, HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1, HUF_DECSIZE = 1 << HUF_DECBITS, HUF_DECMASK = HUF_DECSIZE - 1;
struct HufDec { int len = 0; unsigned lit = 0; std::vector<unsigned> p; };
struct BitIn { const unsigned char* in; const unsigned char* end; unsigned long long c = 0; int lc = 0;
    bool getChar() { if (in >= end) return false; c = (c << 8) | *in++; lc += 8; return true; }
    bool getBits(int n, unsigned& out) { while (lc < n) if (!getChar()) return false; lc -= n; out = (unsigned)((c >> lc) & ((1ull << n) - 1ull)); return true; } };
bool huf_unpack_table(BitIn& b, int im, int iM, std::vector<unsigned long long>& hcode) {
    for (; im <= iM; im++) {
        unsigned l; if (!b.getBits(6, l)) return false;
        hcode[(size_t)im] = l;
        if (l == 63u) { unsigned z; if (!b.getBits(8, z)) return false; int zerun = (int)z + 6; if (im + zerun > iM + 1) return false; while (zerun--) hcode[(size_t)im++] = 0; im--; }
        else if (l >= 59u) { int zerun = (int)l - 59 + 2; if (im + zerun > iM + 1) return false; while (zerun--) hcode[(size_t)im++] = 0; im--; }
    }
    // canonical codes from the lengths (hufCanonicalCodeTable): code = hcode >> 6, length = hcode & 63
    unsigned long long n[59]; for (auto& v : n) v = 0;
    for (int i = 0; i < HUF_ENCSIZE; i++) { if (hcode[(size_t)i] > 58) return false; n[hcode[(size_t)i]] += 1; }
    unsigned long long c = 0;
    for (int i = 58; i > 0; --i) { const unsigned long long nc = (c + n[i]) >> 1; n[i] = c; c = nc; }
    for (int i = 0; i < HUF_ENCSIZE; i++) { const int l = (int)hcode[(size_t)i]; if (l > 0) hcode[(size_t)i] = (unsigned long long)l | (n[l]++ << 6); }
    return true;
}
bool huf_uncompress(const unsigned char* data, size_t nData, std::vector<unsigned short>& out, size_t nRaw) {
    out.clear(); if (nRaw == 0) return true;
    if (nData < 20) return false;
    auto u32 = [&](size_t o) { unsigned v; memcpy(&v, data + o, 4); return v; };
    const unsigned im = u32(0), iM = u32(4), nBits = u32(12);
    if (im >= (unsigned)HUF_ENCSIZE || iM >= (unsigned)HUF_ENCSIZE || im > iM) return false;
    std::vector<unsigned long long> hcode((size_t)HUF_ENCSIZE, 0ull);
    BitIn tb{data + 20, data + nData};
    if (!huf_unpack_table(tb, (int)im, (int)iM, hcode)) return false;
    const unsigned char* ptr = tb.in;
    if ((unsigned long long)nBits > 8ull * (unsigned long long)(data + nData - ptr)) return false;
    std::vector<HufDec> dec((size_t)HUF_DECSIZE);
    for (unsigned s = im; s <= iM; s++) {      // hufBuildDecTable
        const unsigned long long c = hcode[s] >> 6; const int l = (int)(hcode[s] & 63);
        if (l == 0) continue;
        if (c >> l) return false;
        if (l > HUF_DECBITS) { HufDec& pl = dec[(size_t)(c >> (l - HUF_DECBITS))]; if (pl.len) return false; pl.lit++; pl.p.push_back(s); }
        else { size_t first = (size_t)(c << (HUF_DECBITS - l)); for (size_t i = 0; i < ((size_t)1 << (HUF_DECBITS - l)); i++) { HufDec& pl = dec[first + i]; if (pl.len || !pl.p.empty()) return false; pl.len = l; pl.lit = s; } }
    }
    out.reserve(nRaw);
    BitIn b{ptr, ptr + (nBits + 7u) / 8u};
    const unsigned rlc = iM;
    auto getCode = [&](unsigned po) -> bool {
        if (po == rlc) {
            if (b.lc < 8 && !b.getChar()) return false;
            b.lc -= 8; const unsigned cs = (unsigned)((b.c >> b.lc) & 0xFFu);
            if (out.empty() || out.size() + cs > nRaw) return false;
            const unsigned short sv = out.back(); out.insert(out.end(), cs, sv);
        } else { if (out.size() >= nRaw) return false; out.push_back((unsigned short)po); }
        return true;
    };
    while (b.in < b.end) {
        b.getChar();
        while (b.lc >= HUF_DECBITS) {
            const HufDec& pl = dec[(size_t)((b.c >> (b.lc - HUF_DECBITS)) & (unsigned long long)HUF_DECMASK)];
            if (pl.len) { b.lc -= pl.len; if (!getCode(pl.lit)) return false; }
            else {
                if (pl.p.empty()) return false;
                size_t j = 0;
                for (; j < pl.p.size(); j++) {
                    const int l = (int)(hcode[pl.p[j]] & 63);
                    while (b.lc < l && b.in < b.end) b.getChar();
                    if (b.lc >= l && (hcode[pl.p[j]] >> 6) == ((b.c >> (b.lc - l)) & ((1ull << l) - 1ull))) { b.lc -= l; if (!getCode(pl.p[j])) return false; break; }
                }
                if (j == pl.p.size()) return false;
            }
        }
    }
    const int i = (8 - (int)nBits) & 7; b.c >>= i; b.lc -= i;
    while (b.lc > 0) {
        const HufDec& pl = dec[(size_t)((b.c << (HUF_DECBITS - b.lc)) & (unsigned long long)HUF_DECMASK)];
        if (!pl.len || pl.len > b.lc) return false;
        b.lc -= pl.len; if (!getCode(pl.lit)) return false;
    }
    return out.size() == nRaw;
}
inline void wdec14(unsigned short l, unsigned short h, unsigned short& a, unsigned short& b) {
    const short ls = (short)l, hs = (short)h; const int hi = hs, ai = ls + (hi & 1) + (hi >> 1);
    a = (unsigned short)(short)ai; b = (unsigned short)(short)(ai - hi);
}
inline void wdec16(unsigned short l, unsigned short h, unsigned short& a, unsigned short& b) {
    const int m = l, d = h, bb = (m - (d >> 1)) & 0xFFFF, aa = (d + bb - 0x8000) & 0xFFFF;
    b = (unsigned short)bb; a = (unsigned short)aa;
}
void wav2_decode(unsigned short* in, int nx, int ox, int ny, int oy, unsigned short mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx; int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    while (p >= 1) {
        unsigned short* py = in; unsigned short* ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        unsigned short i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            unsigned short* px = py; unsigned short* ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                unsigned short* p01 = px + ox1; unsigned short* p10 = px + oy1; unsigned short* p11 = p10 + ox1;
                if (w14) { wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11); wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11); }
                else { wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11); wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11); }
            }
            if (nx & p) { unsigned short* p10 = px + oy1; if (w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10); *px = i00; }
        }
        if (ny & p) {
            unsigned short* px = py; unsigned short* ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) { unsigned short* p01 = px + ox1; if (w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01); *px = i00; }
        }
        p2 = p; p >>= 1;
    }
}
// one PIZ block -> the block's scan lines in the file's own (uncompressed) layout: per line, channel after channel. words[k]: 16-bit words per pixel of channel k (half 1, float / uint 2)
bool exr_piz_decode(const unsigned char* src, size_t size, size_t w, size_t lines, const std::vector<int>& words, std::vector<unsigned char>& raw) {
    size_t perLine = 0; for (int k : words) perLine += (size_t)k * w;
    const size_t total = perLine * lines;
    if (size < 4) return false;
    unsigned short minNZ, maxNZ; memcpy(&minNZ, src, 2); memcpy(&maxNZ, src + 2, 2);
    std::vector<unsigned char> bitmap(8192, 0); size_t at = 4;
    if (maxNZ >= 8192) return false;
    if (minNZ <= maxNZ) { const size_t n = (size_t)maxNZ - minNZ + 1; if (at + n > size) return false; memcpy(bitmap.data() + minNZ, src + at, n); at += n; }
    std::vector<unsigned short> lut(65536, 0); size_t k = 0;
    for (unsigned i = 0; i < 65536u; i++) if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7u)))) lut[k++] = (unsigned short)i;
    const unsigned short maxValue = (unsigned short)(k - 1);
    if (at + 4 > size) return false;
    int length; memcpy(&length, src + at, 4); at += 4;
    if (length < 0 || at + (size_t)length > size) return false;
    std::vector<unsigned short> tmp;
    if (!huf_uncompress(src + at, (size_t)length, tmp, total)) return false;
    size_t chStart = 0;
    for (int wk : words) { for (int j = 0; j < wk; j++) wav2_decode(tmp.data() + chStart + (size_t)j, (int)w, wk, (int)lines, (int)(w * (size_t)wk), maxValue); chStart += (size_t)wk * w * lines; }
    for (auto& v : tmp) v = lut[v];
    raw.resize(total * 2);
    std::vector<size_t> cur(words.size()); { size_t o = 0; for (size_t c = 0; c < words.size(); c++) { cur[c] = o; o += (size_t)words[c] * w * lines; } }
    unsigned char* o = raw.data();
    for (size_t y = 0; y < lines; y++) for (size_t c = 0; c < words.size(); c++) { const size_t n = (size_t)words[c] * w; memcpy(o, tmp.data() + cur[c], n * 2); cur[c] += n; o += n * 2; }
    return true;
}

int32_t read_exr(const std::vector<unsigned char>& d, uint32_t& W, uint32_t& H, std::vector<float>& rgb) {
    Rd r{d.data(), d.size(), 0, true};
    if (d.size() < 8 || (unsigned)r.i32() != 20000630u) return PT_ERROR_IO;
    const unsigned ver = (unsigned)r.i32();
    if ((ver & 0xFFu) != 2u) return PT_ERROR_UNSUPPORTED;
    if (ver & (0x800u | 0x1000u)) return PT_ERROR_UNSUPPORTED;                     // deep, multi-part
    const bool tiled = (ver & 0x200u) != 0u;                                        // single-part tiled: level (0, 0) is read (the full-resolution image of a mip-mapped / rip-mapped file)
    const size_t maxName = (ver & 0x400u) ? 255 : 31;
    struct Chan { std::string name; int type, xs, ys; };
    std::vector<Chan> ch; int comp = -1, dw[4] = {0, 0, -1, -1}, lineOrder = 0; bool haveDw = false;
    unsigned tileW = 0, tileH = 0; bool haveTiles = false;
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
        else if (name == "tiles" && type == "tiledesc") { tileW = (unsigned)a.i32(); tileH = (unsigned)a.i32(); (void)a.u8(); haveTiles = a.ok; }      // (the mode byte — level and rounding mode — is not needed for level 0)
    }
    if (ch.empty() || comp < 0 || !haveDw) return PT_ERROR_IO;
    if (comp > 4) return PT_ERROR_UNSUPPORTED;                                      // 0 none, 1 RLE, 2 ZIPS, 3 ZIP, 4 PIZ; PXR24 / B44 / DWA are not read
    const long long w = (long long)dw[2] - dw[0] + 1, h = (long long)dw[3] - dw[1] + 1;
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768 || w * h > (1ll << 28)) return PT_ERROR_IO;      // (a damaged data window must not turn into a 12 GB allocation)
    size_t pixelBytes = 0; int idx[3] = {-1, -1, -1}, yIdx = -1; std::vector<size_t> chBytesBefore(ch.size());      // a line of a block: channel after channel, `width` values each
    for (size_t k = 0; k < ch.size(); k++) {
        if (ch[k].xs != 1 || ch[k].ys != 1) return PT_ERROR_UNSUPPORTED;            // sub-sampled (luminance / chroma) channels
        if (ch[k].type < 0 || ch[k].type > 2) return PT_ERROR_IO;
        chBytesBefore[k] = pixelBytes; pixelBytes += (ch[k].type == 1 ? 2u : 4u);
        if (ch[k].name == "R") idx[0] = (int)k; else if (ch[k].name == "G") idx[1] = (int)k; else if (ch[k].name == "B") idx[2] = (int)k; else if (ch[k].name == "Y") yIdx = (int)k;
    }
    if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0) { if (yIdx < 0) return PT_ERROR_UNSUPPORTED; idx[0] = idx[1] = idx[2] = yIdx; }
    for (int k = 0; k < 3; k++) if (ch[(size_t)idx[k]].type == 0) return PT_ERROR_UNSUPPORTED;      // uint channels carry ids, not radiance
    std::vector<int> pizWords; for (auto& c : ch) pizWords.push_back(c.type == 1 ? 1 : 2);
    W = (uint32_t)w; H = (uint32_t)h; rgb.assign((size_t)w * h * 3, 0.f);
    std::vector<unsigned char> raw, tmp;
    // one chunk of pixel data — a block of scan lines or a tile — of bw x lines pixels at (x0, y0): undo the codec, then pick R, G, B out of the channel-planar lines
    auto chunk = [&](const unsigned char* src, size_t size, size_t x0, size_t y0, size_t bw, size_t lines) -> bool {
        const size_t lineBytes = bw * pixelBytes, want = lines * lineBytes;
        if (size == want) raw.assign(src, src + want);                              // stored as is (also what the codecs fall back to when they do not shrink the chunk)
        else if (comp == 0) return false;
        else if (comp == 1) { if (!exr_rle_decode(src, size, tmp, want)) return false; exr_unpredict_interleave(tmp, raw); }
        else if (comp == 4) { if (!exr_piz_decode(src, size, bw, lines, pizWords, raw) || raw.size() != want) return false; }
        else { tmp.resize(want); uLongf got = (uLongf)want; if (uncompress(tmp.data(), &got, src, (uLong)size) != Z_OK || got != want) return false; exr_unpredict_interleave(tmp, raw); }
        for (size_t l = 0; l < lines; l++) {
            const unsigned char* line = raw.data() + l * lineBytes; float* o = &rgb[((y0 + l) * (size_t)w + x0) * 3];
            for (int k = 0; k < 3; k++) {
                const Chan& cc = ch[(size_t)idx[k]]; const unsigned char* p = line + chBytesBefore[(size_t)idx[k]] * bw;
                if (cc.type == 1) for (size_t x = 0; x < bw; x++) { unsigned short v; memcpy(&v, p + 2 * x, 2); o[3 * x + (size_t)k] = half_to_float(v); }
                else for (size_t x = 0; x < bw; x++) { float v; memcpy(&v, p + 4 * x, 4); o[3 * x + (size_t)k] = v; }
            }
        }
        return true;
    };
    (void)lineOrder;                                                                // every chunk carries its own position: the order of the chunks in the file does not matter
    if (tiled) {
        if (!haveTiles || tileW == 0u || tileH == 0u || tileW > 32768u || tileH > 32768u) return PT_ERROR_IO;
        const size_t nx = ((size_t)w + tileW - 1) / tileW, ny = ((size_t)h + tileH - 1) / tileH;
        if (nx * ny * 8 > d.size()) return PT_ERROR_IO;
        std::vector<unsigned long long> offs(nx * ny); for (auto& o : offs) o = r.u64();      // the table's first nx x ny entries are level (0, 0) in every level mode; further levels follow and are not read
        if (!r.ok) return PT_ERROR_IO;
        for (size_t t = 0; t < offs.size(); t++) {
            if (offs[t] > d.size() || d.size() - (size_t)offs[t] < 20) return PT_ERROR_IO;
            Rd c{d.data(), d.size(), (size_t)offs[t], true};
            const long long tx = c.i32(), ty = c.i32(), lx = c.i32(), ly = c.i32(); const int size = c.i32();
            if (!c.ok || size < 0 || c.i + (size_t)size > c.n || lx != 0 || ly != 0 || tx < 0 || ty < 0 || (size_t)tx >= nx || (size_t)ty >= ny) return PT_ERROR_IO;
            const size_t x0 = (size_t)tx * tileW, y0 = (size_t)ty * tileH;
            if (!chunk(d.data() + c.i, (size_t)size, x0, y0, std::min<size_t>(tileW, (size_t)w - x0), std::min<size_t>(tileH, (size_t)h - y0))) return PT_ERROR_IO;
        }
        return PT_OK;
    }
    const unsigned linesPerBlock = comp == 3 ? 16u : (comp == 4 ? 32u : 1u);
    const size_t blocks = ((size_t)h + linesPerBlock - 1) / linesPerBlock;
    if (blocks * 8 > d.size()) return PT_ERROR_IO;                                   // the offset table alone would not fit the file
    std::vector<unsigned long long> offs(blocks); for (auto& o : offs) o = r.u64();
    if (!r.ok) return PT_ERROR_IO;
    for (size_t b = 0; b < blocks; b++) {
        if (offs[b] > d.size() || d.size() - (size_t)offs[b] < 8) return PT_ERROR_IO;
        Rd c{d.data(), d.size(), (size_t)offs[b], true};
        const long long y0 = (long long)c.i32() - dw[1]; const int size = c.i32();
        if (!c.ok || size < 0 || c.i + (size_t)size > c.n || y0 < 0 || y0 >= h) return PT_ERROR_IO;
        if (!chunk(d.data() + c.i, (size_t)size, 0, (size_t)y0, (size_t)w, (size_t)std::min<long long>(linesPerBlock, h - y0))) return PT_ERROR_IO;
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
