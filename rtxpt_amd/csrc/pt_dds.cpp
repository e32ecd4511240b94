// mi355pt — .dds reader (host side). The reference's texture pipeline prefers a .dds next to every .png (MaterialsBaker.cpp:178-191; its compression script writes
// BC7, SampleCommon.cpp:700-730: `nvtt_export -f 23`) and accepts .dds environment maps (Sample.cpp:116); Donut's TextureCache / DDSFile.cpp reads them and the
// texture unit decodes the blocks. Donut is not vendored, so container and block formats are restated from their published definitions (the DDS programming guide
// for the header; the Khronos Data Format Specification for S3TC / RGTC / BPTC) and the decoders are checked against Pillow's independent ones (tests/test_dds.py).
//   BC7 decodes exactly as specified (the format leaves no freedom); BC1-BC3 colour interpolation is the "ideal" (2 c0 + c1 + 1) / 3 rule, which hardware is
//   allowed to approximate; only the top mip level is read (the library builds its own chain, as for .png files).
// Read: legacy FourCC DXT1 / DXT3 / DXT5 / ATI1 / BC4U / ATI2 / BC5U / 113 (RGBA16F) / 116 (RGBA32F), uncompressed 32-bit RGBA / BGRA / BGRX masks, and the DX10 header
// with BC1 / BC2 / BC3 / BC4 / BC5 / BC7 (TYPELESS, UNORM, SRGB), R8G8B8A8, B8G8R8A8, R16G16B16A16_FLOAT, R32G32B32A32_FLOAT, BC6H (UF16 and SF16, all fourteen modes: float RGBA out). Not read (PT_ERROR_UNSUPPORTED): cube maps,
// volumes, arrays, signed BC4 / BC5.
#include "../../include/mi355pt.h"
#include "pt_bcn_tables.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t fourcc(char a, char b, char c, char d) { return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24); }

// ---- S3TC colour block (BC1; the colour half of BC2 / BC3): two RGB565 endpoints, 16 two-bit selectors
void decode_color_block(const uint8_t* b, uint8_t out[16][4], bool bc1) {
    const uint32_t c0 = b[0] | (b[1] << 8), c1 = b[2] | (b[3] << 8);
    uint8_t pal[4][4];
    auto expand = [](uint32_t c, uint8_t* o) { uint32_t r = (c >> 11) & 31u, g = (c >> 5) & 63u, bl = c & 31u; o[0] = (uint8_t)((r << 3) | (r >> 2)); o[1] = (uint8_t)((g << 2) | (g >> 4)); o[2] = (uint8_t)((bl << 3) | (bl >> 2)); o[3] = 255; };
    expand(c0, pal[0]); expand(c1, pal[1]);
    if (c0 > c1 || !bc1) { for (int k = 0; k < 3; k++) { pal[2][k] = (uint8_t)((2 * pal[0][k] + pal[1][k] + 1) / 3); pal[3][k] = (uint8_t)((pal[0][k] + 2 * pal[1][k] + 1) / 3); } pal[2][3] = pal[3][3] = 255; }
    else { for (int k = 0; k < 3; k++) { pal[2][k] = (uint8_t)((pal[0][k] + pal[1][k]) / 2); pal[3][k] = 0; } pal[2][3] = 255; pal[3][3] = 0; }      // three colours + transparent black
    const uint32_t sel = rd32(b + 4);
    for (int i = 0; i < 16; i++) memcpy(out[i], pal[(sel >> (2 * i)) & 3u], 4);
}
// ---- RGTC / BC3-alpha block: two 8-bit endpoints, 16 three-bit selectors
void decode_alpha_block(const uint8_t* b, uint8_t out[16]) {
    const uint32_t a0 = b[0], a1 = b[1]; uint8_t pal[8]; pal[0] = (uint8_t)a0; pal[1] = (uint8_t)a1;
    if (a0 > a1) for (uint32_t k = 1; k < 7; k++) pal[k + 1] = (uint8_t)(((7 - k) * a0 + k * a1 + 3) / 7);
    else { for (uint32_t k = 1; k < 5; k++) pal[k + 1] = (uint8_t)(((5 - k) * a0 + k * a1 + 2) / 5); pal[6] = 0; pal[7] = 255; }
    unsigned long long bits = 0; for (int k = 0; k < 6; k++) bits |= (unsigned long long)b[2 + k] << (8 * k);
    for (int i = 0; i < 16; i++) out[i] = pal[(bits >> (3 * i)) & 7u];
}

// ---- BPTC / BC7
struct BitReader { unsigned __int128 v; uint32_t pos = 0;                      // the 128-bit block, least significant bit first
    explicit BitReader(const uint8_t* p) { unsigned long long lo, hi; memcpy(&lo, p, 8); memcpy(&hi, p + 8, 8); v = ((unsigned __int128)hi << 64) | lo; }
    uint32_t get(uint32_t n) { if (!n) return 0u; uint32_t r = (uint32_t)(v >> pos) & ((1u << n) - 1u); pos += n; return r; } };
struct Bc7Mode { uint8_t subsets, partitionBits, rotationBits, indexSelBits, colorBits, alphaBits, endpointP, sharedP, indexBits, index2Bits; };
const Bc7Mode kModes[8] = {{3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
                           {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0}};
const uint8_t kWeights2[4] = {0, 21, 43, 64}, kWeights3[8] = {0, 9, 18, 27, 37, 46, 55, 64}, kWeights4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};
inline uint32_t bc7_weight(uint32_t bits, uint32_t idx) { return bits == 2 ? kWeights2[idx] : (bits == 3 ? kWeights3[idx] : kWeights4[idx]); }
inline uint8_t bc7_lerp(uint32_t a, uint32_t b, uint32_t w) { return (uint8_t)(((64u - w) * a + w * b + 32u) >> 6); }
void decode_bc7_block(const uint8_t* b, uint8_t out[16][4]) {
    uint32_t mode = 0; while (mode < 8 && !((b[0] >> mode) & 1u)) mode++;
    if (mode == 8) { memset(out, 0, 64); return; }                                     // reserved: decodes to transparent black
    const Bc7Mode& m = kModes[mode]; BitReader r(b); r.pos = mode + 1u;
    const uint32_t partition = r.get(m.partitionBits), rotation = r.get(m.rotationBits), indexSel = r.get(m.indexSelBits);
    uint32_t ep[6][4];                                                                 // endpoint 2 s + e of subset s, channels RGBA
    const uint32_t nEp = 2u * m.subsets;
    for (uint32_t c = 0; c < 3; c++) for (uint32_t e = 0; e < nEp; e++) ep[e][c] = r.get(m.colorBits);
    for (uint32_t e = 0; e < nEp; e++) ep[e][3] = m.alphaBits ? r.get(m.alphaBits) : 255u;
    uint32_t cb = m.colorBits, ab = m.alphaBits;
    if (m.endpointP) { for (uint32_t e = 0; e < nEp; e++) { const uint32_t p = r.get(1); for (uint32_t c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | p; if (ab) ep[e][3] = (ep[e][3] << 1) | p; } cb++; if (ab) ab++; }
    if (m.sharedP) { for (uint32_t s = 0; s < m.subsets; s++) { const uint32_t p = r.get(1); for (uint32_t e = 2 * s; e < 2 * s + 2; e++) for (uint32_t c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | p; } cb++; }
    for (uint32_t e = 0; e < nEp; e++) {                                               // to 8 bits: shift up, replicate the top bits below
        for (uint32_t c = 0; c < 3; c++) { uint32_t v = ep[e][c] << (8u - cb); ep[e][c] = v | (v >> cb); }
        if (ab) { uint32_t v = ep[e][3] << (8u - ab); ep[e][3] = v | (v >> ab); }
    }
    uint32_t subset[16], anchor[3] = {0u, 0u, 0u};
    for (uint32_t i = 0; i < 16; i++) subset[i] = m.subsets == 1 ? 0u : (m.subsets == 2 ? (ptbcn::kPartition2[partition] >> i) & 1u : (ptbcn::kPartition3[partition] >> (2u * i)) & 3u);
    if (m.subsets == 2) anchor[1] = ptbcn::kAnchor2[partition];
    if (m.subsets == 3) { anchor[1] = ptbcn::kAnchor3a[partition]; anchor[2] = ptbcn::kAnchor3b[partition]; }
    uint32_t idx[2][16];
    for (uint32_t i = 0; i < 16; i++) idx[0][i] = r.get(i == anchor[subset[i]] ? m.indexBits - 1u : m.indexBits);
    if (m.index2Bits) for (uint32_t i = 0; i < 16; i++) idx[1][i] = r.get(i == 0u ? m.index2Bits - 1u : m.index2Bits);
    for (uint32_t i = 0; i < 16; i++) {
        const uint32_t* e0 = ep[2u * subset[i]]; const uint32_t* e1 = ep[2u * subset[i] + 1u];
        uint32_t cw, aw;
        if (!m.index2Bits) cw = aw = bc7_weight(m.indexBits, idx[0][i]);
        else if (!indexSel) { cw = bc7_weight(m.indexBits, idx[0][i]); aw = bc7_weight(m.index2Bits, idx[1][i]); }
        else { cw = bc7_weight(m.index2Bits, idx[1][i]); aw = bc7_weight(m.indexBits, idx[0][i]); }
        uint8_t px[4] = {bc7_lerp(e0[0], e1[0], cw), bc7_lerp(e0[1], e1[1], cw), bc7_lerp(e0[2], e1[2], cw), bc7_lerp(e0[3], e1[3], aw)};
        if (rotation == 1u) { uint8_t t = px[3]; px[3] = px[0]; px[0] = t; } else if (rotation == 2u) { uint8_t t = px[3]; px[3] = px[1]; px[1] = t; } else if (rotation == 3u) { uint8_t t = px[3]; px[3] = px[2]; px[2] = t; }
        memcpy(out[i], px, 4);
    }
}

float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u; uint32_t u;
    if (e == 0u) { if (!m) u = s; else { int k = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; k++; } u = s | ((uint32_t)(113 - k) << 23) | ((mm & 1023u) << 13); } }
    else if (e == 31u) u = s | 0x7F800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

enum Kind { K_NONE, K_BC1, K_BC2, K_BC3, K_BC4, K_BC5, K_BC7, K_RGBA8, K_BGRA8, K_BGRX8, K_RGBA16F, K_RGBA32F, K_BC6U, K_BC6S };

// ---- BC6H (BPTC float), unsigned (UF16) and signed (SF16): fourteen modes; the endpoint fields of each mode are scattered over bits 2 / 5 .. 76 (two regions) or .. 64 (one region)
// as listed below in bit order — field names of the published format: r/g/b = channel, w x y z = endpoints 0 1 2 3, "n" or "n-m" = bit(s) of that field, ascending bit positions.
// Two-region modes carry a 5-bit partition at bits 77-81 and 3-bit indices from bit 82, one-region modes 4-bit indices from bit 65; anchor texels have one bit fewer.
// Endpoints 1-3 of the transformed modes are deltas (sign-extended) to endpoint 0, the sum wrapping within its width; unquantisation, interpolation (weights x / 64) and
// the final x 31 / 64 (unsigned) or x 31 / 32 (signed) are the format's. Checked against Pillow's decoder on random blocks of every mode (tests/test_dds.py); the
// two-region modes 7.6 / 9.5 and mode 11 also against the reference's own encoder (pt_envcube.h).
struct Bc6Mode { uint8_t modeBits, modeValue, regions, transformed, prec, dr, dg, db; const char* layout; };
const Bc6Mode kBc6Modes[14] = {
    {2, 0x00, 2, 1, 10, 5, 5, 5, "gy4 by4 bz4 rw0-9 gw0-9 bw0-9 rx0-4 gz4 gy0-3 gx0-4 bz0 gz0-3 bx0-4 bz1 by0-3 ry0-4 bz2 rz0-4 bz3"},
    {2, 0x01, 2, 1, 7, 6, 6, 6, "gy5 gz4 gz5 rw0-6 bz0 bz1 by4 gw0-6 by5 bz2 gy4 bw0-6 bz3 bz5 bz4 rx0-5 gy0-3 gx0-5 gz0-3 bx0-5 by0-3 ry0-5 rz0-5"},
    {5, 0x02, 2, 1, 11, 5, 4, 4, "rw0-9 gw0-9 bw0-9 rx0-4 rw10 gy0-3 gx0-3 gw10 bz0 gz0-3 bx0-3 bw10 bz1 by0-3 ry0-4 bz2 rz0-4 bz3"},
    {5, 0x06, 2, 1, 11, 4, 5, 4, "rw0-9 gw0-9 bw0-9 rx0-3 rw10 gz4 gy0-3 gx0-4 gw10 gz0-3 bx0-3 bw10 bz1 by0-3 ry0-3 bz0 bz2 rz0-3 gy4 bz3"},
    {5, 0x0A, 2, 1, 11, 4, 4, 5, "rw0-9 gw0-9 bw0-9 rx0-3 rw10 by4 gy0-3 gx0-3 gw10 bz0 gz0-3 bx0-4 bw10 by0-3 ry0-3 bz1 bz2 rz0-3 bz4 bz3"},
    {5, 0x0E, 2, 1, 9, 5, 5, 5, "rw0-8 by4 gw0-8 gy4 bw0-8 bz4 rx0-4 gz4 gy0-3 gx0-4 bz0 gz0-3 bx0-4 bz1 by0-3 ry0-4 bz2 rz0-4 bz3"},
    {5, 0x12, 2, 1, 8, 6, 5, 5, "rw0-7 gz4 by4 gw0-7 bz2 gy4 bw0-7 bz3 bz4 rx0-5 gy0-3 gx0-4 bz0 gz0-3 bx0-4 bz1 by0-3 ry0-5 rz0-5"},
    {5, 0x16, 2, 1, 8, 5, 6, 5, "rw0-7 bz0 by4 gw0-7 gy5 gy4 bw0-7 gz5 bz4 rx0-4 gz4 gy0-3 gx0-5 gz0-3 bx0-4 bz1 by0-3 ry0-4 bz2 rz0-4 bz3"},
    {5, 0x1A, 2, 1, 8, 5, 5, 6, "rw0-7 bz1 by4 gw0-7 by5 gy4 bw0-7 bz5 bz4 rx0-4 gz4 gy0-3 gx0-4 bz0 gz0-3 bx0-5 by0-3 ry0-4 bz2 rz0-4 bz3"},
    {5, 0x1E, 2, 0, 6, 6, 6, 6, "rw0-5 gz4 bz0 bz1 by4 gw0-5 gy5 by5 bz2 gy4 bw0-5 gz5 bz3 bz5 bz4 rx0-5 gy0-3 gx0-5 gz0-3 bx0-5 by0-3 ry0-5 rz0-5"},
    {5, 0x03, 1, 0, 10, 10, 10, 10, "rw0-9 gw0-9 bw0-9 rx0-9 gx0-9 bx0-9"},
    {5, 0x07, 1, 1, 11, 9, 9, 9, "rw0-9 gw0-9 bw0-9 rx0-8 rw10 gx0-8 gw10 bx0-8 bw10"},
    {5, 0x0B, 1, 1, 12, 8, 8, 8, "rw0-9 gw0-9 bw0-9 rx0-7 rw11 rw10 gx0-7 gw11 gw10 bx0-7 bw11 bw10"},
    {5, 0x0F, 1, 1, 16, 4, 4, 4, "rw0-9 gw0-9 bw0-9 rx0-3 rw15 rw14 rw13 rw12 rw11 rw10 gx0-3 gw15 gw14 gw13 gw12 gw11 gw10 bx0-3 bw15 bw14 bw13 bw12 bw11 bw10"},
};
inline uint32_t bc6_bit(const uint8_t* b, uint32_t i) { return (b[i >> 3] >> (i & 7u)) & 1u; }
inline int32_t bc6_sext(uint32_t v, uint32_t bits) { return (bits < 32u && (v & (1u << (bits - 1u)))) ? (int32_t)(v | ~((1u << bits) - 1u)) : (int32_t)v; }
inline int32_t bc6_unquantize(int32_t c, uint32_t bits, bool isSigned) {
    if (!isSigned) { if (bits >= 15u) return c; if (c == 0) return 0; if (c == (int32_t)((1u << bits) - 1u)) return 0xFFFF; return (int32_t)((((uint32_t)c << 16) + 0x8000u) >> bits); }
    if (bits >= 16u) return c;
    const bool neg = c < 0; uint32_t a = (uint32_t)(neg ? -c : c), u;
    if (a == 0u) u = 0u; else if (a >= (1u << (bits - 1u)) - 1u) u = 0x7FFFu; else u = ((a << 15) + 0x4000u) >> (bits - 1u);
    return neg ? -(int32_t)u : (int32_t)u;
}
// one 16-byte block -> 16 texels of 3 half-float bit patterns (a reserved mode decodes to zero, as the format prescribes)
void decode_bc6h_block(const uint8_t* b, bool isSigned, uint16_t out[16][3]) {
    memset(out, 0, sizeof(uint16_t) * 48);
    const uint32_t m2 = b[0] & 3u, m5 = b[0] & 31u; const Bc6Mode* M = nullptr;
    for (const Bc6Mode& k : kBc6Modes) if ((k.modeBits == 2 && k.modeValue == m2) || (k.modeBits == 5 && m2 >= 2u && k.modeValue == m5)) { M = &k; break; }
    if (!M) return;
    int32_t e[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    uint32_t pos = M->modeBits;
    for (const char* s = M->layout; *s;) {                                        // "<channel><endpoint><lo>[-<hi>]"
        while (*s == ' ') s++; if (!*s) break;
        const int ch = *s == 'r' ? 0 : (*s == 'g' ? 1 : 2); s++;
        const int ep = *s == 'w' ? 0 : (*s == 'x' ? 1 : (*s == 'y' ? 2 : 3)); s++;
        uint32_t lo = 0; while (*s >= '0' && *s <= '9') lo = lo * 10u + (uint32_t)(*s++ - '0');
        uint32_t hi = lo; if (*s == '-') { s++; hi = 0; while (*s >= '0' && *s <= '9') hi = hi * 10u + (uint32_t)(*s++ - '0'); }
        for (uint32_t j = lo; j <= hi; j++) e[ep][ch] |= (int32_t)(bc6_bit(b, pos++) << j);
    }
    const uint32_t P = M->prec, D[3] = {M->dr, M->dg, M->db}, nEp = M->regions * 2u;
    for (int c = 0; c < 3; c++) {
        if (isSigned) e[0][c] = bc6_sext((uint32_t)e[0][c], P);
        for (uint32_t k = 1; k < nEp; k++) {
            if (M->transformed) { const int32_t v = (e[0][c] + bc6_sext((uint32_t)e[k][c], D[c])) & (int32_t)((1u << P) - 1u); e[k][c] = isSigned ? bc6_sext((uint32_t)v, P) : v; }
            else if (isSigned) e[k][c] = bc6_sext((uint32_t)e[k][c], D[c]);
        }
        for (uint32_t k = 0; k < nEp; k++) e[k][c] = bc6_unquantize(e[k][c], P, isSigned);
    }
    static const uint8_t w3[8] = {0, 9, 18, 27, 37, 46, 55, 64}, w4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};
    uint32_t partition = 0, ipos = 65u, anchor = 0u;
    if (M->regions == 2) { for (uint32_t j = 0; j < 5u; j++) partition |= bc6_bit(b, 77u + j) << j; ipos = 82u; anchor = ptbcn::kAnchor2[partition]; }
    for (uint32_t i = 0; i < 16u; i++) {
        const uint32_t region = M->regions == 2 ? (ptbcn::kPartition2[partition] >> i) & 1u : 0u;
        uint32_t nb = M->regions == 2 ? 3u : 4u; if (i == 0u || (M->regions == 2 && i == anchor)) nb--;
        uint32_t idx = 0; for (uint32_t j = 0; j < nb; j++) idx |= bc6_bit(b, ipos++) << j;
        const int32_t wt = M->regions == 2 ? w3[idx] : w4[idx];
        for (int c = 0; c < 3; c++) {
            const int32_t a = e[2u * region][c], bb = e[2u * region + 1u][c], v = (a * (64 - wt) + bb * wt + 32) >> 6;
            if (!isSigned) out[i][c] = (uint16_t)((v * 31) >> 6);
            else { const int32_t f = v < 0 ? -(((-v) * 31) >> 5) : (v * 31) >> 5; out[i][c] = f < 0 ? (uint16_t)(0x8000u | (uint32_t)(-f)) : (uint16_t)f; }
        }
    }
}
// one level of a float-capable format (RGBA16F / RGBA32F / BC6H) as RGBA32F texels, alpha 1 for the block format
enum FloatKind { FK_RGBA16F, FK_RGBA32F, FK_BC6U, FK_BC6S };
static size_t float_level_bytes(FloatKind k, size_t w, size_t h) { return k == FK_RGBA16F ? w * h * 8u : (k == FK_RGBA32F ? w * h * 16u : ((w + 3u) / 4u) * ((h + 3u) / 4u) * 16u); }
static void decode_float_level(FloatKind k, const uint8_t* src, size_t w, size_t h, float* out) {
    const size_t npx = w * h;
    if (k == FK_RGBA32F) { memcpy(out, src, npx * 16u); return; }
    if (k == FK_RGBA16F) { for (size_t i = 0; i < npx * 4u; i++) out[i] = half_to_float((uint16_t)(src[2 * i] | (src[2 * i + 1] << 8))); return; }
    const size_t bw = (w + 3u) / 4u, bh = (h + 3u) / 4u;
    for (size_t by = 0; by < bh; by++) for (size_t bx = 0; bx < bw; bx++) {
        uint16_t hb[16][3]; decode_bc6h_block(src + (by * bw + bx) * 16u, k == FK_BC6S, hb);
        for (uint32_t y = 0; y < 4u; y++) for (uint32_t x = 0; x < 4u; x++) { const size_t X = bx * 4u + x, Y = by * 4u + y; if (X >= w || Y >= h) continue;
            float* o = out + (Y * w + X) * 4u; for (int c = 0; c < 3; c++) o[c] = half_to_float(hb[y * 4u + x][c]); o[3] = 1.0f; }
    }
}
struct Bytes { const uint8_t* p; size_t n; size_t size() const { return n; } const uint8_t* data() const { return p; } const uint8_t& operator[](size_t i) const { return p[i]; } };
int32_t read_dds_bytes(const Bytes& d, uint32_t* width, uint32_t* height, uint32_t* format, void** pixels);
int32_t read_dds(const char* path, uint32_t* width, uint32_t* height, uint32_t* format, void** pixels) {
    if (!path || !width || !height || !format || !pixels) return PT_ERROR_INVALID_ARGUMENT;
    *pixels = nullptr; *width = *height = 0; *format = PT_TEX_RGBA8_UNORM;
    FILE* f = fopen(path, "rb"); if (!f) return PT_ERROR_IO;
    std::vector<uint8_t> d; { uint8_t buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) { d.insert(d.end(), buf, buf + n); if (d.size() > ((size_t)1 << 31)) { fclose(f); return PT_ERROR_IO; } } fclose(f); }
    return read_dds_bytes(Bytes{d.data(), d.size()}, width, height, format, pixels);
}
int32_t read_dds_bytes(const Bytes& d, uint32_t* width, uint32_t* height, uint32_t* format, void** pixels) {
    if (!d.p || !width || !height || !format || !pixels) return PT_ERROR_INVALID_ARGUMENT;
    *pixels = nullptr; *width = *height = 0; *format = PT_TEX_RGBA8_UNORM;
    if (d.size() < 128 || memcmp(d.data(), "DDS ", 4) != 0 || rd32(&d[4]) != 124u || rd32(&d[76]) != 32u) return PT_ERROR_IO;
    const uint32_t h = rd32(&d[12]), w = rd32(&d[16]), depth = rd32(&d[24]), hflags = rd32(&d[8]), pfFlags = rd32(&d[80]), cc = rd32(&d[84]), bitCount = rd32(&d[88]);
    const uint32_t rmask = rd32(&d[92]), gmask = rd32(&d[96]), bmask = rd32(&d[100]), amask = rd32(&d[104]), caps2 = rd32(&d[112]);
    if (w == 0 || h == 0 || w > 32768u || h > 32768u) return PT_ERROR_IO;
    if ((caps2 & 0x200u) || ((hflags & 0x800000u) && depth > 1u) || (caps2 & 0x200000u)) return PT_ERROR_UNSUPPORTED;          // cube map / volume
    size_t off = 128; Kind k = K_NONE; bool srgb = false;
    if ((pfFlags & 0x4u) && cc == fourcc('D', 'X', '1', '0')) {
        if (d.size() < 148) return PT_ERROR_IO;
        const uint32_t dxgi = rd32(&d[128]), dim = rd32(&d[132]), misc = rd32(&d[136]), arr = rd32(&d[140]); off = 148;
        if (dim != 3u || (misc & 0x4u) || arr > 1u) return PT_ERROR_UNSUPPORTED;          // only 2D, no cube, no array
        switch (dxgi) {
        case 70: case 71: k = K_BC1; break; case 72: k = K_BC1; srgb = true; break; case 73: case 74: k = K_BC2; break; case 75: k = K_BC2; srgb = true; break;
        case 76: case 77: k = K_BC3; break; case 78: k = K_BC3; srgb = true; break; case 79: case 80: k = K_BC4; break; case 82: case 83: k = K_BC5; break;
        case 97: case 98: k = K_BC7; break; case 99: k = K_BC7; srgb = true; break;
        case 27: case 28: k = K_RGBA8; break; case 29: k = K_RGBA8; srgb = true; break; case 87: k = K_BGRA8; break; case 91: k = K_BGRA8; srgb = true; break;
        case 10: k = K_RGBA16F; break; case 2: k = K_RGBA32F; break;
        case 94: case 95: k = K_BC6U; break; case 96: k = K_BC6S; break;                   // BC6H_TYPELESS / _UF16, _SF16
        default: return PT_ERROR_UNSUPPORTED;                                             // signed RGTC (81, 84), everything else
        }
    } else if (pfFlags & 0x4u) {
        if (cc == fourcc('D', 'X', 'T', '1')) k = K_BC1; else if (cc == fourcc('D', 'X', 'T', '3') || cc == fourcc('D', 'X', 'T', '2')) k = K_BC2;
        else if (cc == fourcc('D', 'X', 'T', '5') || cc == fourcc('D', 'X', 'T', '4')) k = K_BC3;
        else if (cc == fourcc('A', 'T', 'I', '1') || cc == fourcc('B', 'C', '4', 'U')) k = K_BC4; else if (cc == fourcc('A', 'T', 'I', '2') || cc == fourcc('B', 'C', '5', 'U')) k = K_BC5;
        else if (cc == 113u) k = K_RGBA16F; else if (cc == 116u) k = K_RGBA32F; else return PT_ERROR_UNSUPPORTED;
    } else if ((pfFlags & 0x40u) && bitCount == 32u) {
        if (rmask == 0xFFu && gmask == 0xFF00u && bmask == 0xFF0000u) k = (pfFlags & 0x1u) && amask == 0xFF000000u ? K_RGBA8 : K_NONE;
        else if (rmask == 0xFF0000u && gmask == 0xFF00u && bmask == 0xFFu) k = (pfFlags & 0x1u) && amask == 0xFF000000u ? K_BGRA8 : K_BGRX8;
        if (k == K_NONE) return PT_ERROR_UNSUPPORTED;
    } else return PT_ERROR_UNSUPPORTED;
    const size_t bw = (w + 3u) / 4u, bh = (h + 3u) / 4u, npx = (size_t)w * h;
    size_t need = 0;
    switch (k) { case K_BC1: case K_BC4: need = bw * bh * 8u; break; case K_BC2: case K_BC3: case K_BC5: case K_BC7: case K_BC6U: case K_BC6S: need = bw * bh * 16u; break;
                 case K_RGBA8: case K_BGRA8: case K_BGRX8: need = npx * 4u; break; case K_RGBA16F: need = npx * 8u; break; case K_RGBA32F: need = npx * 16u; break; default: break; }
    if (d.size() < off + need) return PT_ERROR_IO;
    const uint8_t* src = d.data() + off;
    if (k == K_RGBA16F || k == K_RGBA32F || k == K_BC6U || k == K_BC6S) {      // float texels out (HDR blocks: alpha 1)
        float* out = (float*)malloc(npx * 16u); if (!out) return PT_ERROR_IO;
        decode_float_level(k == K_RGBA16F ? FK_RGBA16F : (k == K_RGBA32F ? FK_RGBA32F : (k == K_BC6U ? FK_BC6U : FK_BC6S)), src, w, h, out);
        *pixels = out; *format = PT_TEX_RGBA32F; *width = w; *height = h; return PT_OK;
    }
    uint8_t* out = (uint8_t*)malloc(npx * 4u); if (!out) return PT_ERROR_IO;
    if (k == K_RGBA8) memcpy(out, src, npx * 4u);
    else if (k == K_BGRA8 || k == K_BGRX8) for (size_t i = 0; i < npx; i++) { out[4 * i] = src[4 * i + 2]; out[4 * i + 1] = src[4 * i + 1]; out[4 * i + 2] = src[4 * i]; out[4 * i + 3] = k == K_BGRA8 ? src[4 * i + 3] : 255; }
    else {
        const size_t blockBytes = (k == K_BC1 || k == K_BC4) ? 8u : 16u;
        for (size_t by = 0; by < bh; by++) for (size_t bx = 0; bx < bw; bx++) {
            const uint8_t* blk = src + (by * bw + bx) * blockBytes; uint8_t px[16][4];
            if (k == K_BC1) decode_color_block(blk, px, true);
            else if (k == K_BC2) { decode_color_block(blk + 8, px, false); for (int i = 0; i < 16; i++) { uint32_t a = (blk[i >> 1] >> ((i & 1) * 4)) & 15u; px[i][3] = (uint8_t)(a * 17u); } }
            else if (k == K_BC3) { uint8_t a[16]; decode_color_block(blk + 8, px, false); decode_alpha_block(blk, a); for (int i = 0; i < 16; i++) px[i][3] = a[i]; }
            else if (k == K_BC4) { uint8_t a[16]; decode_alpha_block(blk, a); for (int i = 0; i < 16; i++) { px[i][0] = a[i]; px[i][1] = px[i][2] = 0; px[i][3] = 255; } }
            else if (k == K_BC5) { uint8_t a[16], g[16]; decode_alpha_block(blk, a); decode_alpha_block(blk + 8, g); for (int i = 0; i < 16; i++) { px[i][0] = a[i]; px[i][1] = g[i]; px[i][2] = 0; px[i][3] = 255; } }
            else decode_bc7_block(blk, px);
            for (uint32_t y = 0; y < 4u; y++) for (uint32_t x = 0; x < 4u; x++) { const size_t X = bx * 4u + x, Y = by * 4u + y; if (X < w && Y < h) memcpy(out + (Y * w + X) * 4u, px[y * 4u + x], 4); }
        }
    }
    *pixels = out; *format = srgb ? PT_TEX_RGBA8_SRGB : PT_TEX_RGBA8_UNORM; *width = w; *height = h;
    return PT_OK;
}

// A cube-map .dds as an environment SOURCE (EnvMapBaker.cpp:399-411: a loaded texture with arraySize 6 becomes m_loadedSourceBackgroundTextureCubemap, BackgroundSourceType 2):
// all six faces (legacy DDSCAPS2_CUBEMAP_ALLFACES or a DX10 header with the TEXTURECUBE flag), float-capable formats only, the top level of every face.
int32_t read_dds_cube_bytes(const Bytes& d, uint32_t* dim, float** rgba) {
    if (!d.p || !dim || !rgba) return PT_ERROR_INVALID_ARGUMENT;
    *rgba = nullptr; *dim = 0;
    if (d.size() < 128 || memcmp(d.data(), "DDS ", 4) != 0 || rd32(&d[4]) != 124u || rd32(&d[76]) != 32u) return PT_ERROR_IO;
    const uint32_t h = rd32(&d[12]), w = rd32(&d[16]), hflags = rd32(&d[8]), pfFlags = rd32(&d[80]), cc = rd32(&d[84]), caps2 = rd32(&d[112]);
    uint32_t mips = (hflags & 0x20000u) ? rd32(&d[28]) : 1u; if (mips == 0u) mips = 1u;
    if (w == 0 || w != h || w > 16384u || mips > 15u) return (w != h) ? PT_ERROR_UNSUPPORTED : PT_ERROR_IO;
    size_t off = 128; FloatKind k; bool cube = (caps2 & 0x200u) != 0u;
    if (cube && (caps2 & 0xFC00u) != 0xFC00u) return PT_ERROR_UNSUPPORTED;                                   // a partial cube (not all six faces)
    if ((pfFlags & 0x4u) && cc == fourcc('D', 'X', '1', '0')) {
        if (d.size() < 148) return PT_ERROR_IO;
        const uint32_t dxgi = rd32(&d[128]), rdim = rd32(&d[132]), misc = rd32(&d[136]), arr = rd32(&d[140]); off = 148;
        if (rdim != 3u || !(misc & 0x4u) || arr != 1u) return PT_ERROR_UNSUPPORTED;                           // one cube: 2D resource, TEXTURECUBE, array size 1 (x 6 faces)
        cube = true;
        switch (dxgi) { case 10: k = FK_RGBA16F; break; case 2: k = FK_RGBA32F; break; case 94: case 95: k = FK_BC6U; break; case 96: k = FK_BC6S; break; default: return PT_ERROR_UNSUPPORTED; }
    } else if (pfFlags & 0x4u) { if (cc == 113u) k = FK_RGBA16F; else if (cc == 116u) k = FK_RGBA32F; else return PT_ERROR_UNSUPPORTED; }
    else return PT_ERROR_UNSUPPORTED;
    if (!cube) return PT_ERROR_UNSUPPORTED;                                                                   // a 2D file: pt_image_read_dds / pt_image_read_float
    size_t faceBytes = 0; for (uint32_t l = 0; l < mips; l++) { const size_t lw = (w >> l) ? (w >> l) : 1u; faceBytes += float_level_bytes(k, lw, lw); }
    if (d.size() < off + 6u * faceBytes) return PT_ERROR_IO;
    const size_t npx = (size_t)w * w;
    float* out = (float*)malloc(6u * npx * 16u); if (!out) return PT_ERROR_IO;
    for (uint32_t f = 0; f < 6u; f++) decode_float_level(k, d.data() + off + f * faceBytes, w, w, out + f * npx * 4u);      // faces in the file's (= D3D's) order: +X -X +Y -Y +Z -Z
    *rgba = out; *dim = w;
    return PT_OK;
}

} // namespace

extern "C" int32_t pt_image_read_dds_cube(const char* path, uint32_t* dim, float** rgbaFaces) {
    if (!path || !dim || !rgbaFaces) return PT_ERROR_INVALID_ARGUMENT;
    *rgbaFaces = nullptr; *dim = 0;
    try {
        FILE* f = fopen(path, "rb"); if (!f) return PT_ERROR_IO;
        std::vector<uint8_t> d; { uint8_t buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) { d.insert(d.end(), buf, buf + n); if (d.size() > ((size_t)1 << 32)) { fclose(f); return PT_ERROR_IO; } } fclose(f); }
        return read_dds_cube_bytes(Bytes{d.data(), d.size()}, dim, rgbaFaces);
    } catch (...) { *rgbaFaces = nullptr; return PT_ERROR_IO; }
}
extern "C" int32_t pt_image_read_dds_memory(const void* bytes, size_t size, uint32_t* width, uint32_t* height, uint32_t* format, void** pixels) {
    try { return read_dds_bytes(Bytes{(const uint8_t*)bytes, size}, width, height, format, pixels); } catch (...) { if (pixels) *pixels = nullptr; return PT_ERROR_IO; }
}
extern "C" int32_t pt_image_read_dds(const char* path, uint32_t* width, uint32_t* height, uint32_t* format, void** pixels) {
    try { return read_dds(path, width, height, format, pixels); } catch (...) { if (pixels) *pixels = nullptr; return PT_ERROR_IO; }
}
