// mi355pt — JPEG reader (host side) for glTF images. Donut's TextureCache hands .jpg files to stb_image (not vendored); the stream format is ITU-T T.81 and the
// sample reconstruction chosen here is the IJG reference decoder's default path (jidctint "islow" inverse DCT, "fancy" triangle up-sampling, its fixed-point YCbCr
// -> RGB tables), which tests/test_jpeg.py checks against Pillow (libjpeg-turbo) — stb_image's own integer IDCT and resampler differ from it by a unit here and
// there, as every pair of conforming decoders may. Read: baseline / extended-sequential / progressive Huffman streams, 8 bits per sample, greyscale or three
// components (YCbCr, or RGB when an Adobe marker says so), sampling factors 1 or 2, restart intervals. Not read: arithmetic coding, lossless, hierarchical,
// 12-bit, CMYK.
#include "../../include/mi355pt.h"
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ptjpeg {
namespace {

const uint8_t kZigZag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff { bool present = false; uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int32_t mincode[17], maxcode[18], valptr[17];
    void build() { int32_t code = 0, k = 0; for (int l = 1; l <= 16; l++) { valptr[l] = k; mincode[l] = code; code += bits[l]; k += bits[l]; maxcode[l] = bits[l] ? code - 1 : -1; code <<= 1; } maxcode[17] = 0x7FFFFFFF; } };
struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int bw = 0, bh = 0;      // blocks per row / column (padded to whole MCUs)
              int dw = 0, dh = 0;                                                         // down-sampled size in samples (what the up-sampler treats as real)
              std::vector<int16_t> coef; std::vector<uint8_t> plane; };

struct Decoder {
    const uint8_t* p; const uint8_t* end;
    uint16_t qt[4][64]; bool qtPresent[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart = 0;
    bool progressive = false, adobe = false; int adobeTransform = -1;
    Comp comp[3];
    // entropy-coded segment state
    uint32_t bitbuf = 0; int bitcnt = 0; bool hitMarker = false; int eobrun = 0;

    bool fail = false;
    int byte() { return p < end ? *p++ : (fail = true, 0); }
    int word() { int a = byte(); return (a << 8) | byte(); }

    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hitMarker && p < end) {
                b = *p;
                if (b == 0xFF) { int n = p + 1 < end ? p[1] : 0xD9; if (n == 0) { p += 2; } else { hitMarker = true; b = 0; } }      // a marker ends the segment: feed zeros
                else p++;
            } else hitMarker = true;
            bitbuf |= (uint32_t)b << (24 - bitcnt); bitcnt += 8;
        }
    }
    int getbits(int n) { if (n <= 0) return 0; if (n > 16) { fail = true; return 0; }      // (a size category beyond 16 bits only comes out of a damaged table)
                         if (bitcnt < n) fill(); int v = (int)(bitbuf >> (32 - n)); bitbuf <<= n; bitcnt -= n; return v; }
    int getbit() { return getbits(1); }
    static int extend(int v, int n) { return (n < 1 || n > 16) ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }
    int decode(const Huff& h) {
        int32_t code = 0;
        for (int l = 1; l <= 16; l++) { code = (code << 1) | getbit(); if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + (code - h.mincode[l])]; }
        fail = true; return 0;
    }
    void reset_entropy() { bitbuf = 0; bitcnt = 0; hitMarker = false; eobrun = 0; }

    bool parse_tables_until_sos(int& marker);
    bool decode_scan();
    void finish(std::vector<uint8_t>& rgba);
};

bool read_markers_and_scans(Decoder& d, std::vector<uint8_t>& rgba) {
    if (d.word() != 0xFFD8) return false;
    bool sawSOF = false, sawScan = false;
    for (;;) {
        int m;
        do { m = d.byte(); if (d.fail) return sawScan; } while (m != 0xFF);
        do { m = d.byte(); if (d.fail) return sawScan; } while (m == 0xFF);
        if (m == 0xD9) break;
        if (m == 0x00 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        int len = d.word(); if (d.fail || len < 2 || d.p + (len - 2) > d.end) return sawScan;
        const uint8_t* seg = d.p; const uint8_t* segEnd = d.p + (len - 2);
        if (m == 0xDB) {                                             // DQT
            while (seg < segEnd) { int pq = *seg >> 4, tq = *seg & 15; seg++; if (tq > 3 || pq > 1 || seg + (pq ? 128 : 64) > segEnd) return false;
                for (int i = 0; i < 64; i++) { int v = pq ? ((seg[0] << 8) | seg[1]) : seg[0]; seg += pq ? 2 : 1; d.qt[tq][kZigZag[i]] = (uint16_t)v; } d.qtPresent[tq] = true; }
        } else if (m == 0xC4) {                                      // DHT
            while (seg < segEnd) { int tc = *seg >> 4, th = *seg & 15; seg++; if (tc > 1 || th > 3 || seg + 16 > segEnd) return false;
                Huff& h = tc ? d.ac[th] : d.dc[th]; int total = 0; for (int l = 1; l <= 16; l++) { h.bits[l] = seg[l - 1]; total += h.bits[l]; } seg += 16;
                if (total > 256 || seg + total > segEnd) return false; memcpy(h.vals, seg, (size_t)total); seg += total; h.present = true; h.build(); }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {            // SOF0 / SOF1 / SOF2
            if (sawSOF || len < 8) return false; sawSOF = true; d.progressive = m == 0xC2;
            int prec = seg[0]; d.height = (seg[1] << 8) | seg[2]; d.width = (seg[3] << 8) | seg[4]; d.ncomp = seg[5];
            if (prec != 8 || d.width <= 0 || d.height <= 0 || d.width > 32768 || d.height > 32768 || (d.ncomp != 1 && d.ncomp != 3) || len < 8 + 3 * d.ncomp) return false;
            if ((size_t)d.width * (size_t)d.height / 1024u > (size_t)(d.end - d.p) + 65536u) return false;      // a header that promises far more pixels than the stream could hold (even a flat image needs a few bits per block): no giant allocations for a tiny file
            for (int c = 0; c < d.ncomp; c++) { Comp& k = d.comp[c]; k.id = seg[6 + 3 * c]; k.h = seg[7 + 3 * c] >> 4; k.v = seg[7 + 3 * c] & 15; k.tq = seg[8 + 3 * c];
                if (k.h < 1 || k.h > 2 || k.v < 1 || k.v > 2 || k.tq > 3) return false; if (k.h > d.hmax) d.hmax = k.h; if (k.v > d.vmax) d.vmax = k.v; }
            if (d.ncomp == 1) { d.comp[0].h = d.comp[0].v = 1; d.hmax = d.vmax = 1; }
            d.mcux = (d.width + 8 * d.hmax - 1) / (8 * d.hmax); d.mcuy = (d.height + 8 * d.vmax - 1) / (8 * d.vmax);
            for (int c = 0; c < d.ncomp; c++) { Comp& k = d.comp[c]; k.bw = d.mcux * k.h; k.bh = d.mcuy * k.v; k.dw = (d.width * k.h + d.hmax - 1) / d.hmax; k.dh = (d.height * k.v + d.vmax - 1) / d.vmax;
                if ((size_t)k.bw * k.bh > ((size_t)1 << 24)) return false; k.coef.assign((size_t)k.bw * k.bh * 64, 0); }
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) return false;      // lossless / hierarchical / arithmetic
        else if (m == 0xDD) { if (len < 4) return false; d.restart = (seg[0] << 8) | seg[1]; }
        else if (m == 0xEE && len >= 14 && !memcmp(seg, "Adobe", 5)) { d.adobe = true; d.adobeTransform = seg[11]; }
        else if (m == 0xDA) {                                        // SOS: header, then the entropy-coded data
            if (!sawSOF) return false;
            d.p = seg;      // decode_scan parses the header itself
            if (!d.decode_scan()) return false;
            sawScan = true; continue;
        }
        d.p = segEnd;
    }
    return sawScan;
}

bool Decoder::decode_scan() {
    int ns = byte(); if (ns < 1 || ns > ncomp) return false;
    int order[3];
    for (int i = 0; i < ns; i++) { int id = byte(), t = byte(), k = -1; for (int c = 0; c < ncomp; c++) if (comp[c].id == id) k = c; if (k < 0) return false; order[i] = k; comp[k].td = t >> 4; comp[k].ta = t & 15; if (comp[k].td > 3 || comp[k].ta > 3) return false; }
    int ss = byte(), se = byte(), a = byte(), ah = a >> 4, al = a & 15;
    if (fail) return false;
    if (!progressive) { ss = 0; se = 63; ah = al = 0; }
    else if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13 || ah > 13) return false;
    for (int i = 0; i < ns; i++) { const Comp& k = comp[order[i]]; if ((ss == 0 && !(progressive && ah) && !dc[k.td].present) || (se > 0 && !ac[k.ta].present)) return false; }
    reset_entropy();
    int pred[3] = {0, 0, 0};
    // block (bx, by) of component k
    auto block = [&](Comp& k, int bx, int by) -> int16_t* { return &k.coef[((size_t)by * k.bw + bx) * 64]; };
    auto decode_block = [&](Comp& k, int16_t* b, int ci) {
        if (!progressive) {
            int t = decode(dc[k.td]); int diff = (t > 0 && t <= 16) ? extend(getbits(t), t) : 0; pred[ci] = (int)((unsigned)pred[ci] + (unsigned)diff); b[0] = (int16_t)pred[ci];
            for (int i = 1; i < 64;) { int rs = decode(ac[k.ta]), r = rs >> 4, s = rs & 15; if (!s) { if (r != 15) break; i += 16; continue; } i += r; if (i > 63) { fail = true; break; } b[kZigZag[i]] = (int16_t)extend(getbits(s), s); i++; }
            return;
        }
        if (ss == 0) {                                               // DC scan
            if (!ah) { int t = decode(dc[k.td]); int diff = (t > 0 && t <= 16) ? extend(getbits(t), t) : 0; pred[ci] = (int)((unsigned)pred[ci] + (unsigned)diff); b[0] = (int16_t)((unsigned)pred[ci] << al); }
            else if (getbit()) b[0] = (int16_t)(b[0] | (1 << al));
            return;
        }
        if (!ah) {                                                   // AC first pass
            if (eobrun > 0) { eobrun--; return; }
            for (int i = ss; i <= se;) { int rs = decode(ac[k.ta]), r = rs >> 4, s = rs & 15;
                if (!s) { if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += getbits(r); break; } i += 16; continue; }
                i += r; if (i > 63) { fail = true; break; } b[kZigZag[i]] = (int16_t)((unsigned)extend(getbits(s), s) << al); i++; }
            return;
        }
        // AC refinement
        const int p1 = 1 << al, m1 = -(1 << al); int i = ss;
        if (eobrun <= 0) {
            for (; i <= se;) {
                int rs = decode(ac[k.ta]), r = rs >> 4, s = rs & 15, val = 0;
                if (s) { if (s != 1) { fail = true; return; } val = getbit() ? p1 : m1; }
                else if (r < 15) { eobrun = 1 << r; if (r) eobrun += getbits(r); break; }
                for (; i <= se; i++) { int16_t& c = b[kZigZag[i]];
                    if (c != 0) { if (getbit() && !(c & p1)) c = (int16_t)(c >= 0 ? c + p1 : c + m1); }
                    else { if (r == 0) { if (val) c = (int16_t)val; i++; break; } r--; } }
                if (fail) return;
            }
        }
        if (eobrun > 0) { for (; i <= se; i++) { int16_t& c = b[kZigZag[i]]; if (c != 0 && getbit() && !(c & p1)) c = (int16_t)(c >= 0 ? c + p1 : c + m1); } eobrun--; }
    };
    int unitsX, unitsY; Comp* single = ns == 1 ? &comp[order[0]] : nullptr;
    if (single) { unitsX = (single->dw + 7) / 8; unitsY = (single->dh + 7) / 8; } else { unitsX = mcux; unitsY = mcuy; }      // a one-component scan walks that component's own blocks
    int count = 0;
    for (int uy = 0; uy < unitsY; uy++) for (int ux = 0; ux < unitsX; ux++) {
        if (restart && count && count % restart == 0) {              // RSTn: byte-align, skip the marker, reset predictors
            reset_entropy(); pred[0] = pred[1] = pred[2] = 0;
            while (p < end && !(p[0] == 0xFF && p + 1 < end && p[1] >= 0xD0 && p[1] <= 0xD7)) { if (p[0] == 0xFF && p + 1 < end && p[1] != 0 && p[1] != 0xFF) break; p++; }
            if (p + 1 < end && p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) p += 2;
        }
        if (single) decode_block(*single, block(*single, ux, uy), 0);
        else for (int i = 0; i < ns; i++) { Comp& k = comp[order[i]]; for (int y = 0; y < k.v; y++) for (int x = 0; x < k.h; x++) decode_block(k, block(k, ux * k.h + x, uy * k.v + y), i); }
        if (fail) return false;
        count++;
    }
    // leave p at the next marker
    if (!hitMarker) { while (p < end && !(p[0] == 0xFF && p + 1 < end && p[1] != 0 && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++; }
    return true;
}

// jidctint.c "islow": Loeffler-Ligtenberg-Moschytz, 13-bit constants, two extra bits kept through the column pass
typedef long long i64;      // the butterflies in 64 bits: identical to the 32-bit reference arithmetic on every valid stream, defined behaviour on hostile ones
inline uint8_t clamp8(i64 v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
    const int C_BITS = 13, P1 = 2;
    const i64 F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    i64 ws[64], dq[64];
    for (int k = 0; k < 64; k++) { long v = (long)in[k] * (long)q[k]; dq[k] = v < -32767 ? -32767 : (v > 32767 ? 32767 : v); }      // (a valid 8-bit stream stays far inside; a hostile one must not overflow the fixed-point butterflies)
    for (int c = 0; c < 8; c++) {
        const i64* i = dq + c; i64* w = ws + c;
        if (!i[8] && !i[16] && !i[24] && !i[32] && !i[40] && !i[48] && !i[56]) { i64 dcv = i[0] * (1 << P1); for (int r = 0; r < 8; r++) w[8 * r] = dcv; continue; }
        i64 z2 = i[16], z3 = i[48];
        i64 z1 = (z2 + z3) * F0_541, t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
        z2 = i[0]; z3 = i[32];
        i64 t0 = (z2 + z3) * (1 << C_BITS), t1 = (z2 - z3) * (1 << C_BITS);
        i64 t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = i[56]; t1 = i[40]; t2 = i[24]; t3 = i[8];
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; i64 z4 = t1 + t3, z5 = (z3 + z4) * F1_175;
        t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501; z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5; t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        const int sh = C_BITS - P1; const i64 rnd = 1ll << (sh - 1);
        w[0] = (t10 + t3 + rnd) >> sh; w[56] = (t10 - t3 + rnd) >> sh; w[8] = (t11 + t2 + rnd) >> sh; w[48] = (t11 - t2 + rnd) >> sh;
        w[16] = (t12 + t1 + rnd) >> sh; w[40] = (t12 - t1 + rnd) >> sh; w[24] = (t13 + t0 + rnd) >> sh; w[32] = (t13 - t0 + rnd) >> sh;
    }
    for (int r = 0; r < 8; r++) {
        const i64* w = ws + 8 * r; uint8_t* o = out + r * stride;
        const int sh = C_BITS + P1 + 3; const i64 rnd = 1ll << (sh - 1);
        i64 z2 = w[2], z3 = w[6];
        i64 z1 = (z2 + z3) * F0_541, t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
        i64 t0 = (w[0] + w[4]) * (1 << C_BITS), t1 = (w[0] - w[4]) * (1 << C_BITS);
        i64 t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = w[7]; t1 = w[5]; t2 = w[3]; t3 = w[1];
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; i64 z4 = t1 + t3, z5 = (z3 + z4) * F1_175;
        t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501; z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5; t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        o[0] = clamp8(((t10 + t3 + rnd) >> sh) + 128); o[7] = clamp8(((t10 - t3 + rnd) >> sh) + 128); o[1] = clamp8(((t11 + t2 + rnd) >> sh) + 128); o[6] = clamp8(((t11 - t2 + rnd) >> sh) + 128);
        o[2] = clamp8(((t12 + t1 + rnd) >> sh) + 128); o[5] = clamp8(((t12 - t1 + rnd) >> sh) + 128); o[3] = clamp8(((t13 + t0 + rnd) >> sh) + 128); o[4] = clamp8(((t13 - t0 + rnd) >> sh) + 128);
    }
}

void Decoder::finish(std::vector<uint8_t>& rgba) {
    for (int c = 0; c < ncomp; c++) { Comp& k = comp[c]; const int stride = k.bw * 8; k.plane.assign((size_t)stride * k.bh * 8, 0);
        for (int by = 0; by < k.bh; by++) for (int bx = 0; bx < k.bw; bx++) idct_islow(&k.coef[((size_t)by * k.bw + bx) * 64], qt[k.tq], &k.plane[(size_t)by * 8 * stride + bx * 8], stride); }
    // up-sample every component to the full size: jdsample.c's "fancy" triangle filters where a factor is 2 (and the component is wider than 2 samples), replication otherwise
    std::vector<std::vector<uint8_t>> full((size_t)ncomp);
    for (int c = 0; c < ncomp; c++) {
        Comp& k = comp[c]; const int stride = k.bw * 8; const int fx = hmax / k.h, fy = vmax / k.v; std::vector<uint8_t>& o = full[(size_t)c]; o.assign((size_t)width * height, 0);
        auto S = [&](int x, int y) -> int { if (y < 0) y = 0; if (y >= k.dh) y = k.dh - 1; return k.plane[(size_t)y * stride + x]; };
        const bool fancyH = fx == 2 && k.dw > 2, fancy = fancyH || (fy == 2 && fx == 1);
        for (int y = 0; y < height; y++) {
            uint8_t* row = &o[(size_t)y * width];
            if (fx == 1 && fy == 1) { memcpy(row, &k.plane[(size_t)y * stride], (size_t)width); continue; }
            if (!fancy || (fx == 2 && !fancyH)) { for (int x = 0; x < width; x++) row[x] = (uint8_t)S(x / fx, y / fy); continue; }
            const int sy = y / fy;
            if (fy == 1) {                                           // h2v1: 3/4 nearer + 1/4 further, bias 1 on the left output, 2 on the right
                for (int x = 0; x < width; x++) { const int sx = x >> 1, v = S(sx, sy);
                    if (!(x & 1)) row[x] = sx == 0 ? (uint8_t)v : (uint8_t)((3 * v + S(sx - 1, sy) + 1) >> 2);
                    else row[x] = sx == k.dw - 1 ? (uint8_t)v : (uint8_t)((3 * v + S(sx + 1, sy) + 2) >> 2); }
            } else if (fx == 1) {                                    // h1v2: bias 1 on the upper output row, 2 on the lower
                const int other = (y & 1) ? sy + 1 : sy - 1, bias = (y & 1) ? 2 : 1;
                for (int x = 0; x < width; x++) row[x] = (uint8_t)((3 * S(x, sy) + S(x, other) + bias) >> 2);
            } else {                                                 // h2v2: column sums 3 near + far, then 3/4 - 1/4 across with biases 8 / 7
                const int other = (y & 1) ? sy + 1 : sy - 1;
                auto col = [&](int sx) { return 3 * S(sx, sy) + S(sx, other); };
                for (int x = 0; x < width; x++) { const int sx = x >> 1, t = col(sx);
                    if (!(x & 1)) row[x] = sx == 0 ? (uint8_t)((t * 4 + 8) >> 4) : (uint8_t)((3 * t + col(sx - 1) + 8) >> 4);
                    else row[x] = sx == k.dw - 1 ? (uint8_t)((t * 4 + 7) >> 4) : (uint8_t)((3 * t + col(sx + 1) + 7) >> 4); }
            }
        }
    }
    rgba.resize((size_t)width * height * 4);
    const bool ycc = ncomp == 3 && !(adobe && adobeTransform == 0);
    for (size_t i = 0; i < (size_t)width * height; i++) {
        uint8_t* o = &rgba[4 * i]; o[3] = 255;
        if (ncomp == 1) { o[0] = o[1] = o[2] = full[0][i]; continue; }
        const int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
        if (!ycc) { o[0] = (uint8_t)y; o[1] = full[1][i]; o[2] = full[2][i]; continue; }
        // jdcolor.c: 16-bit fixed point, ONE_HALF folded into the Cr terms
        o[0] = clamp8(y + ((91881 * cr + 32768) >> 16));
        o[1] = clamp8(y + ((-22554 * cb - 46802 * cr + 32768) >> 16));
        o[2] = clamp8(y + ((116130 * cb + 32768) >> 16));
    }
}

} // namespace

bool decode(const uint8_t* data, size_t size, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    Decoder d; d.p = data; d.end = data + size; memset(d.qt, 0, sizeof d.qt);
    if (!read_markers_and_scans(d, rgba) || d.width <= 0 || d.height <= 0) return false;
    for (int c = 0; c < d.ncomp; c++) if (!d.qtPresent[d.comp[c].tq]) return false;
    if (d.ncomp == 3 && !(d.comp[0].h == d.hmax && d.comp[0].v == d.vmax)) return false;          // (chroma finer than luma: not a layout any encoder in use writes)
    d.finish(rgba); w = (uint32_t)d.width; h = (uint32_t)d.height;
    return true;
}

} // namespace ptjpeg

extern "C" int32_t pt_image_read_jpeg(const void* bytes, size_t size, uint32_t* width, uint32_t* height, void** rgba8) {
    if (!bytes || !width || !height || !rgba8) return PT_ERROR_INVALID_ARGUMENT;
    *rgba8 = nullptr; *width = *height = 0;
    try {
        std::vector<uint8_t> px; uint32_t w = 0, h = 0;
        if (!ptjpeg::decode((const uint8_t*)bytes, size, w, h, px)) return PT_ERROR_IO;
        void* out = malloc(px.size()); if (!out) return PT_ERROR_IO;
        memcpy(out, px.data(), px.size()); *rgba8 = out; *width = w; *height = h;
        return PT_OK;
    } catch (...) { return PT_ERROR_IO; }
}
