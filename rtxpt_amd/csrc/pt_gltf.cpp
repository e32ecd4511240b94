// mi355pt — glTF 2.0 scene import behind pt_load_scene_gltf (Sample::LoadScene + SceneLoaded, Rtxpt/Sample.cpp:447-560;
// material import mirrors MaterialsBaker::ImportFromDonut + PTMaterial::FillData, Rtxpt/Materials/MaterialsBaker.cpp:516-591, 660-705).
// The reference delegates glTF parsing to Donut/cgltf (absent); this is a self-contained reader: JSON, external/base64
// buffers, accessors (float / normalised integer), node hierarchy (matrix or TRS), PNG (all colour types, bit depths, Adam7) via zlib, JPEG and .dds textures.
// Output goes through the same raw-buffer entry points the bakers use (pt_set_materials / pt_set_geometry / pt_set_instances).
#include "../../include/mi355pt.h"
#include <zlib.h>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

struct PxGuard { void* p; ~PxGuard() { if (p) pt_image_free((float*)p); } };      // frees a decoder's buffer even when the copy out of it throws

// ---------------------------------------------------------------- minimal JSON
struct JValue {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    double num = 0; bool b = false; std::string str; std::vector<JValue> arr; std::vector<std::pair<std::string, JValue>> obj;
    const JValue* get(const char* k) const { if (type != Obj) return nullptr; for (auto& kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
    double numOr(const char* k, double d) const { const JValue* v = get(k); return (v && v->type == Num) ? v->num : d; }
    int intOr(const char* k, int d) const { const JValue* v = get(k); return (v && v->type == Num) ? (int)v->num : d; }
    std::string strOr(const char* k, const char* d) const { const JValue* v = get(k); return (v && v->type == Str) ? v->str : std::string(d); }
    size_t size() const { return type == Arr ? arr.size() : 0; }
};
struct JParser {
    const char* p; const char* e; bool ok = true; int depth = 0;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
    bool lit(const char* w, size_t n) { if ((size_t)(e - p) < n || memcmp(p, w, n)) return false; p += n; return true; }
    JValue parse() {
        ws(); JValue v;
        if (p >= e) { ok = false; return v; }
        if (*p == '{' || *p == '[') {
            if (++depth > 512) { ok = false; return v; }          // nesting cap: a "[[[[..." document must not overflow the stack
            if (*p == '{') { p++; v.type = JValue::Obj; ws(); if (p < e && *p == '}') { p++; depth--; return v; }
                while (ok) { ws(); JValue k = parse(); if (!ok || k.type != JValue::Str) { ok = false; break; } ws(); if (p >= e || *p != ':') { ok = false; break; } p++;
                    JValue val = parse(); v.obj.emplace_back(k.str, std::move(val)); ws(); if (p < e && *p == ',') { p++; continue; } if (p < e && *p == '}') { p++; break; } ok = false; }
            } else { p++; v.type = JValue::Arr; ws(); if (p < e && *p == ']') { p++; depth--; return v; }
                while (ok) { v.arr.push_back(parse()); ws(); if (p < e && *p == ',') { p++; continue; } if (p < e && *p == ']') { p++; break; } ok = false; } }
            depth--; return v; }
        if (*p == '"') { p++; v.type = JValue::Str;
            while (p < e && *p != '"') { if (*p == '\\' && p + 1 < e) { p++; char c = *p; if (c == 'n') v.str += '\n'; else if (c == 't') v.str += '\t'; else if (c == 'u') { if (e - p < 5) { ok = false; return v; } p += 4; v.str += '?'; } else v.str += c; p++; } else v.str += *p++; }
            if (p < e) p++; else ok = false; return v; }
        if (lit("true", 4)) { v.type = JValue::Bool; v.b = true; return v; }
        if (lit("false", 5)) { v.type = JValue::Bool; v.b = false; return v; }
        if (lit("null", 4)) return v;
        // a number: copy its characters (at most 63) so that strtod never reads past the buffer, which need not be NUL-terminated
        char buf[64]; size_t n = 0;
        while (p + n < e && n < 63 && (isdigit((unsigned char)p[n]) || p[n] == '-' || p[n] == '+' || p[n] == '.' || p[n] == 'e' || p[n] == 'E')) { buf[n] = p[n]; n++; }
        buf[n] = 0;
        char* end = nullptr; v.num = strtod(buf, &end); if (end == buf) { ok = false; return v; } p += (end - buf); v.type = JValue::Num; return v;
    }
};

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0); size_t r = n > 0 ? fread(out.data(), 1, (size_t)n, f) : 0; fclose(f);
    return r == out.size();
}
bool base64_decode(const std::string& s, size_t start, std::vector<uint8_t>& out) {
    auto val = [](char c) -> int { if (c >= 'A' && c <= 'Z') return c - 'A'; if (c >= 'a' && c <= 'z') return c - 'a' + 26; if (c >= '0' && c <= '9') return c - '0' + 52; if (c == '+') return 62; if (c == '/') return 63; return -1; };
    uint32_t acc = 0; int bits = 0;
    for (size_t i = start; i < s.size(); i++) { int v = val(s[i]); if (v < 0) continue; acc = (acc << 6) | (uint32_t)v; bits += 6; if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xFF)); } }
    return true;
}
bool load_uri(const std::string& baseDir, const std::string& uri, std::vector<uint8_t>& out) {
    if (uri.compare(0, 5, "data:") == 0) { size_t c = uri.find(','); if (c == std::string::npos) return false; return base64_decode(uri, c + 1, out); }
    return read_file(baseDir + uri, out);
}

// ---------------------------------------------------------------- PNG (every colour type and bit depth of the specification, Adam7 interlacing; output RGBA8)
bool decode_png(const std::vector<uint8_t>& d, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 8 || memcmp(d.data(), sig, 8)) return false;
    size_t p = 8; std::vector<uint8_t> idat, plte, trns; int depth = 0, ctype = 0, interlace = 0; w = h = 0;
    auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
    while (p + 8 <= d.size()) {
        uint32_t len = be32(p); std::string type((const char*)&d[p + 4], 4); p += 8;
        if (p + len + 4 > d.size()) return false;
        if (type == "IHDR") { if (len != 13) return false; w = be32(p); h = be32(p + 4); depth = d[p + 8]; ctype = d[p + 9]; interlace = d[p + 12]; }
        else if (type == "PLTE") plte.assign(d.begin() + p, d.begin() + p + len);
        else if (type == "tRNS") trns.assign(d.begin() + p, d.begin() + p + len);
        else if (type == "IDAT") idat.insert(idat.end(), d.begin() + p, d.begin() + p + len);
        else if (type == "IEND") break;
        p += len + 4;
    }
    if (!w || !h || interlace > 1) return false;
    if (w > 32768u || h > 32768u) return false;            // (16 mip levels; also keeps every size below in range)
    int ch = (ctype == 0) ? 1 : (ctype == 2) ? 3 : (ctype == 3) ? 1 : (ctype == 4) ? 2 : (ctype == 6) ? 4 : 0;
    if (!ch) return false;
    // bit depths per colour type (PNG specification, table 11.1): grey 1 2 4 8 16, RGB 8 16, palette 1 2 4 8, grey + alpha 8 16, RGBA 8 16. 16-bit samples are reduced to
    // their high byte (what stb_image hands Donut for an 8-bit texture request); sub-byte grey levels are scaled to 0..255.
    const bool depthOk = depth == 8 || (depth == 16 && ctype != 3) || ((depth == 1 || depth == 2 || depth == 4) && (ctype == 0 || ctype == 3));
    if (!depthOk) return false;
    const size_t bitsPerPixel = (size_t)ch * depth, filterBpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
    // passes: the whole image, or Adam7's seven reduced images (xStart, yStart, xStep, yStep), each filtered on its own
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    size_t rawSize = 0; const int nPass = interlace ? 7 : 1;
    auto passDims = [&](int p, uint32_t& pw, uint32_t& ph) { if (!interlace) { pw = w; ph = h; return; } pw = (w > (uint32_t)adam7[p][0]) ? (w - adam7[p][0] + adam7[p][2] - 1) / adam7[p][2] : 0; ph = (h > (uint32_t)adam7[p][1]) ? (h - adam7[p][1] + adam7[p][3] - 1) / adam7[p][3] : 0; };
    for (int p = 0; p < nPass; p++) { uint32_t pw, ph; passDims(p, pw, ph); if (pw && ph) rawSize += (((size_t)pw * bitsPerPixel + 7) / 8 + 1) * ph; }
    if (rawSize / 1032u > idat.size() + 64u) return false;      // deflate expands at most 1032 : 1: a header that promises more than the stream can hold is damaged (no 16 GB allocation for a 100-byte file)
    std::vector<uint8_t> raw(rawSize);
    uLongf outLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outLen, idat.data(), (uLong)idat.size()) != Z_OK || outLen != raw.size()) return false;
    std::vector<uint16_t> img((size_t)w * h * ch);          // samples as stored (indices, grey levels, 16-bit values): the colour key of tRNS compares these
    size_t rp = 0; std::vector<uint8_t> cur, prev;
    for (int p = 0; p < nPass; p++) {
        uint32_t pw, ph; passDims(p, pw, ph); if (!pw || !ph) continue;
        const size_t stride = ((size_t)pw * bitsPerPixel + 7) / 8; cur.assign(stride, 0); prev.assign(stride, 0);
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t* in = &raw[rp]; const uint8_t ft = in[0]; in++; rp += stride + 1;
            for (size_t x = 0; x < stride; x++) {
                int a = x >= filterBpp ? cur[x - filterBpp] : 0, b = prev[x], c = x >= filterBpp ? prev[x - filterBpp] : 0, v = in[x];
                switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break;
                    case 4: { int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); } break; default: return false; }
                cur[x] = (uint8_t)v;
            }
            const uint32_t oy = interlace ? (uint32_t)adam7[p][1] + y * adam7[p][3] : y;
            for (uint32_t x = 0; x < pw; x++) {
                const uint32_t ox = interlace ? (uint32_t)adam7[p][0] + x * adam7[p][2] : x; uint16_t* o = &img[((size_t)oy * w + ox) * ch];
                for (int k = 0; k < ch; k++) {
                    const size_t si = (size_t)x * ch + k;
                    if (depth == 8) o[k] = cur[si]; else if (depth == 16) o[k] = (uint16_t)((cur[2 * si] << 8) | cur[2 * si + 1]);
                    else { const size_t bit = si * depth; o[k] = (uint16_t)((cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
                }
            }
            cur.swap(prev);
        }
    }
    auto to8 = [&](uint16_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : (depth == 8 ? (uint8_t)v : (uint8_t)(v * 255 / ((1 << depth) - 1))); };
    auto key = [&](size_t o) -> uint16_t { return o + 1 < trns.size() ? (uint16_t)((trns[o] << 8) | trns[o + 1]) : 0; };
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        uint8_t r, g, b, a = 255; const uint16_t* s = &img[i * ch];
        if (ctype == 0) { r = g = b = to8(s[0]); if (trns.size() >= 2 && s[0] == key(0)) a = 0; }
        else if (ctype == 2) { r = to8(s[0]); g = to8(s[1]); b = to8(s[2]); if (trns.size() >= 6 && s[0] == key(0) && s[1] == key(2) && s[2] == key(4)) a = 0; }
        else if (ctype == 4) { r = g = b = to8(s[0]); a = to8(s[1]); }
        else if (ctype == 6) { r = to8(s[0]); g = to8(s[1]); b = to8(s[2]); a = to8(s[3]); }
        else { size_t k = s[0]; if (k * 3 + 2 >= plte.size()) return false; r = plte[k * 3]; g = plte[k * 3 + 1]; b = plte[k * 3 + 2]; if (k < trns.size()) a = trns[k]; }
        rgba[i * 4] = r; rgba[i * 4 + 1] = g; rgba[i * 4 + 2] = b; rgba[i * 4 + 3] = a;
    }
    return true;
}

// ---------------------------------------------------------------- matrices (column-major 4x4 like glTF)
struct M4 { double m[16]; };
M4 m4_identity() { M4 r; memset(&r, 0, sizeof(r)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1; return r; }
M4 m4_mul(const M4& a, const M4& b) { M4 r; for (int c = 0; c < 4; c++) for (int rr = 0; rr < 4; rr++) { double s = 0; for (int k = 0; k < 4; k++) s += a.m[k * 4 + rr] * b.m[c * 4 + k]; r.m[c * 4 + rr] = s; } return r; }
bool m4_inverse(const M4& a, M4& out) {          // general 4 x 4 inverse (column major), cofactor expansion; false for a singular matrix
    const double* m = a.m; double inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (!(det != 0.0) || !std::isfinite(det)) return false;
    for (int i = 0; i < 16; i++) out.m[i] = inv[i] / det;
    return true;
}
M4 m4_trs(const double t[3], const double q[4], const double s[3]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    M4 r = m4_identity();
    r.m[0] = (1 - 2 * (y * y + z * z)) * s[0]; r.m[1] = (2 * (x * y + z * w)) * s[0]; r.m[2] = (2 * (x * z - y * w)) * s[0];
    r.m[4] = (2 * (x * y - z * w)) * s[1]; r.m[5] = (1 - 2 * (x * x + z * z)) * s[1]; r.m[6] = (2 * (y * z + x * w)) * s[1];
    r.m[8] = (2 * (x * z + y * w)) * s[2]; r.m[9] = (2 * (y * z - x * w)) * s[2]; r.m[10] = (1 - 2 * (x * x + y * y)) * s[2];
    r.m[12] = t[0]; r.m[13] = t[1]; r.m[14] = t[2];
    return r;
}

uint32_t pack_snorm8(const float* v, int n) {   // Packing.hlsli:127-140
    uint32_t out = 0;
    for (int i = 0; i < n; i++) { float c = v[i] < -1.f ? -1.f : (v[i] > 1.f ? 1.f : v[i]); out |= ((uint32_t)((int)(c * 127.0f)) & 0xFFu) << (8 * i); }
    return out;
}
uint32_t pack_texture_word(uint32_t index, uint32_t w, uint32_t h) {   // MaterialsBaker.cpp:497-508
    uint32_t mips = 1; { uint32_t m = w > h ? w : h; while (m > 1) { m >>= 1; mips++; } }
    uint32_t baseLOD = (uint32_t)lround(log2((double)w * (double)h));
    return (baseLOD << 24) | (mips << 16) | (index & 0xFFFF);
}

// ---------------------------------------------------------------- RTXPT .material.json -> PTMaterialData
// PTMaterial defaults (MaterialsBaker.h:126-193), Read (MaterialsBaker.cpp:150-259: a missing key keeps the default) and FillData (:516-591)
struct PTMaterialHost {
    float BaseOrDiffuseColor[3] = {1, 1, 1}, SpecularColor[3] = {0, 0, 0}, EmissiveColor[3] = {0, 0, 0};
    float EmissiveIntensity = 1.f, Metalness = 0.f, Roughness = 0.f, Opacity = 1.f, TransmissionFactor = 0.f, DiffuseTransmissionFactor = 0.f, NormalTextureScale = 1.f, IoR = 1.5f;
    bool UseSpecularGlossModel = false, EnableBaseTexture = true, EnableOcclusionRoughnessMetallicTexture = true, EnableNormalTexture = true, EnableEmissiveTexture = true,
         EnableTransmissionTexture = true, EnableAlphaTesting = false, EnableTransmission = false, MetalnessInRedChannel = false, ThinSurface = false, ExcludeFromNEE = false,
         PSDExclude = true, EnableAsAnalyticLightProxy = false, IgnoreMeshTangentSpace = false, UseDonutEmissiveIntensity = false, SkipRender = false;
    float AlphaCutoff = 0.5f; int PSDDominantDeltaLobe = -1, PSDBlockMotionVectorsAtSurfaceType = 0, NestedPriority = 14;
    float VolumeAttenuationDistance = 3.402823466e+38f, VolumeAttenuationColor[3] = {1, 1, 1}, ShadowNoLFadeout = 0.f;
};
// PTMaterial::FillData (MaterialsBaker.cpp:516-591) + GetBindlessTextureIndex: shared by the .material.json path and the glTF import
void fill_data(const PTMaterialHost& m, const bool loaded[5], const uint32_t textureWords[5], PTMaterialData* out) {
    memset(out, 0, sizeof(*out));
    uint32_t f = 0;
    if (m.UseSpecularGlossModel) f |= 0x00000001u;
    if (loaded[0] && m.EnableBaseTexture) f |= 0x00000008u;
    if (loaded[1] && m.EnableOcclusionRoughnessMetallicTexture) f |= 0x00000004u;
    if (loaded[3] && m.EnableEmissiveTexture) f |= 0x00000010u;
    if (loaded[2] && m.EnableNormalTexture) f |= 0x00000020u;
    if (loaded[4] && m.EnableTransmissionTexture && m.EnableTransmission) f |= 0x00000080u;
    if (m.MetalnessInRedChannel) f |= 0x00000100u;
    if (m.ThinSurface || !m.EnableTransmission) f |= 0x00000200u;          // no transmission => thin surface
    if (m.PSDExclude) f |= 0x00000400u;
    if (m.PSDBlockMotionVectorsAtSurfaceType % 2) f |= (1u << 13);
    if (m.PSDBlockMotionVectorsAtSurfaceType / 2) f |= (1u << 14);
    if (m.EnableAsAnalyticLightProxy) f |= 0x00000800u;
    if (m.IgnoreMeshTangentSpace) f |= (1u << 12);
    for (int i = 0; i < 3; i++) { out->BaseOrDiffuseColor[i] = m.BaseOrDiffuseColor[i]; out->SpecularColor[i] = m.SpecularColor[i]; out->EmissiveColor[i] = m.EmissiveColor[i] * m.EmissiveIntensity;
                                  out->AttenuationColor[i] = m.VolumeAttenuationColor[i]; }
    out->Roughness = m.Roughness; out->Metalness = m.Metalness; out->NormalTextureScale = m.NormalTextureScale;
    out->TransmissionFactor = m.EnableTransmission ? m.TransmissionFactor : 0.f; out->DiffuseTransmissionFactor = m.EnableTransmission ? m.DiffuseTransmissionFactor : 0.f;
    out->Opacity = m.Opacity; out->AlphaCutoff = m.AlphaCutoff; out->IoR = m.IoR; out->AttenuationDistance = m.VolumeAttenuationDistance;
    auto texWord = [&](int t, uint32_t bit) -> uint32_t { if (!(f & bit) || !loaded[t]) { f &= ~bit; return 0xFFFFFFFFu; } return textureWords[t]; };      // GetBindlessTextureIndex
    out->BaseOrDiffuseTextureIndex = texWord(0, 0x00000008u); out->MetalRoughOrSpecularTextureIndex = texWord(1, 0x00000004u); out->EmissiveTextureIndex = texWord(3, 0x00000010u);
    out->NormalTextureIndex = texWord(2, 0x00000020u); out->TransmissionTextureIndex = texWord(4, 0x00000080u); out->OcclusionTextureIndex = 0;
    int np = m.NestedPriority < 14 ? m.NestedPriority : 14; if (np < 0) np = 0;
    f |= (uint32_t)np << 28;
    int lobe = m.PSDDominantDeltaLobe + 1; lobe = lobe < 0 ? 0 : (lobe > 7 ? 7 : lobe);
    f |= (uint32_t)lobe << 24;
    out->Flags = f;
    out->ShadowNoLFadeout = m.ShadowNoLFadeout < 0.f ? 0.f : (m.ShadowNoLFadeout > 0.25f ? 0.25f : m.ShadowNoLFadeout);
    out->_padding0 = 42; out->_padding1 = 42.f;       // (the reference writes 42 into both padding words)
}

// a file without a usable "scenes" entry: its roots are the nodes that are nobody's child (visiting every node as a root would pose a child twice — once under its
// parent, once with its local transform only — and the second visit would win the recorded world matrices of joints and mesh nodes)
static std::vector<int> gltf_parentless_nodes(const JValue* nodes) {
    std::vector<int> roots; if (!nodes) return roots;
    std::vector<char> hasParent(nodes->size(), 0);
    for (const JValue& n : nodes->arr) if (const JValue* ch = n.get("children")) for (const JValue& c : ch->arr) { const long k = (long)c.num; if (k >= 0 && (size_t)k < hasParent.size()) hasParent[(size_t)k] = 1; }
    for (size_t i = 0; i < hasParent.size(); i++) if (!hasParent[i]) roots.push_back((int)i);
    return roots;
}
struct Loader {
    pt_context* ctx; std::string baseDir; JValue root; std::vector<std::vector<uint8_t>> buffers; std::string err;
    std::vector<uint32_t> indices; std::vector<float> positions, uvs; std::vector<uint32_t> normals, tangents;
    std::vector<PtGeometryDesc> geoms; std::vector<PtMeshDesc> meshes; std::vector<PtInstanceDesc> instances; std::vector<PTMaterialData> materials;
    std::vector<std::vector<uint8_t>> texPixels; std::vector<PtTextureDesc> texDescs; std::map<std::pair<int, int>, uint32_t> texCache;   // (image, srgb) -> texture word
    std::vector<int> meshMap; std::vector<M4> instanceWorld;      // instanceWorld: the double-precision local-to-world of each instance (scene-graph import composes in double)
    std::vector<std::string> instancePath;                         // the names of the glTF nodes from a scene root down to the instance's node, '/'-separated (what Donut's SceneGraph::FindNode walks)
    std::vector<float> normalsF, tangentsF;                        // the bind-pose NORMAL (3 per vertex) / TANGENT (4 per vertex) as floats: read by pt_gltf_animation_normals
    std::vector<uint16_t> joints; std::vector<float> weights;      // JOINTS_0 / WEIGHTS_0, four per vertex (zeros for an unskinned primitive): read by pt_gltf_animation_positions
    struct SkinnedInstance { int node, skin, mesh; }; std::vector<SkinnedInstance> skinned; std::vector<M4> nodeWorld; bool recordWorlds = false;      // filled by visit() when recordWorlds
    // morph targets (glTF 2.0 3.7.2.2): per geometry the POSITION displacements of its targets (count x numVertices x 3 floats from `first`), per imported mesh its default weights;
    // meshNodes: every (node, mesh) pair visit() met while recordWorlds — read by pt_gltf_animation_positions
    struct GeomMorph { size_t first; uint32_t count; }; std::vector<GeomMorph> geomMorph; std::vector<float> morphDeltas, morphNormalDeltas, morphTangentDeltas; std::vector<std::vector<double>> meshWeights;
    struct MeshNode { int node, mesh; }; std::vector<MeshNode> meshNodes;
    // KHR_lights_punctual (visit()): Donut's glTF importer turns the extension's lights into the scene-graph leaves RTXPT then bakes (DirectionalLight -> the environment cube,
    // Sample.cpp:1361-1388; PointLight / SpotLight -> LightsBaker's analytic lights through PointLightEx / SpotLightEx, Rtxpt/SampleCommon/ExtendedScene.cpp:54-143). Donut is not
    // vendored: the mapping is the extension's own definition over Donut's light members as the .scene.json leaves use them (UNPINNED) — color, intensity (point / spot: luminous
    // intensity -> `intensity`; directional: illuminance -> `irradiance`), spot cone angles in degrees, radius 0, angular size 0; the light points along its node's -Z.
    struct PunctualRaw { int type; float color[3], intensity, inner, outer; M4 world; };      // type 0 point, 1 spot, 2 directional; world: the light node's local-to-world inside this file
    std::vector<PunctualRaw> punctual; uint32_t punctualDropped = 0;
    void punctual_light(int index, const M4& world) {
        const JValue* rext = root.get("extensions"); const JValue* kl = rext ? rext->get("KHR_lights_punctual") : nullptr; const JValue* lights = kl ? kl->get("lights") : nullptr;
        if (!lights || index < 0 || (size_t)index >= lights->size()) return;
        const JValue& l = lights->arr[index]; const std::string type = l.strOr("type", "");
        PunctualRaw r; r.type = type == "directional" ? 2 : (type == "spot" ? 1 : (type == "point" ? 0 : -1)); if (r.type < 0) return;
        r.color[0] = r.color[1] = r.color[2] = 1.f; if (const JValue* c = l.get("color")) if (c->size() == 3) for (int i = 0; i < 3; i++) r.color[i] = (float)c->arr[i].num;
        r.intensity = (float)l.numOr("intensity", 1.0); r.inner = 180.f; r.outer = 180.f; r.world = world;
        const float cx = r.color[0] * r.intensity, cy = r.color[1] * r.intensity, cz = r.color[2] * r.intensity;
        if (sqrtf(cx * cx + cy * cy + cz * cz) <= 1e-7f) { punctualDropped++; return; }      // Sample.cpp:567-573: invisible lights are dropped
        if (r.type == 1) { const JValue* sp = l.get("spot"); r.inner = (float)((sp ? sp->numOr("innerConeAngle", 0.0) : 0.0) * (180.0 / 3.14159265358979323846)); r.outer = (float)((sp ? sp->numOr("outerConeAngle", 0.78539816339744831) : 0.78539816339744831) * (180.0 / 3.14159265358979323846)); }
        punctual.push_back(r);
    }
    // A glTF file's own cameras (glTF 2.0 3.10): Donut's importer hangs a PerspectiveCamera leaf {verticalFov = yfov, zNear = znear} on the node (un-vendored: restated, UNPINNED), and
    // the application lists it with the graph's other cameras (Sample.cpp:580-597). An orthographic camera is not a PerspectiveCamera and is not listed here.
    struct CameraRaw { float yfov, znear; M4 world; std::string name; };
    std::vector<CameraRaw> cameras;
    void gltf_camera(int index, const M4& world, const std::string& nodeName) {
        const JValue* cams = root.get("cameras"); if (!cams || index < 0 || (size_t)index >= cams->size()) return;
        const JValue& c = cams->arr[index]; const JValue* pp = c.get("perspective");
        if (c.strOr("type", "") != "perspective" || !pp) return;
        CameraRaw r; r.yfov = (float)pp->numOr("yfov", 1.0); r.znear = (float)pp->numOr("znear", 1.0); r.world = world; r.name = c.strOr("name", nodeName.c_str());
        cameras.push_back(r);
    }
    // Sample::UpdateCameraFromScene's inputs (Sample.cpp:457-463) of a camera under `parent`: position, row 2 (view direction) and row 1 (up) of scaling(1, 1, -1) * localToWorld
    static PtSceneCameraDesc emit_camera(const CameraRaw& c, const M4& parent) {
        const M4 w = m4_mul(parent, c.world);
        PtSceneCameraDesc d; memset(&d, 0, sizeof(d)); d.verticalFov = c.yfov; d.zNear = c.znear; d.exposureMask = 0x80000000u;      // a plain PerspectiveCamera: no exposure keys to apply
        for (int i = 0; i < 3; i++) { d.position[i] = (float)w.m[12 + i]; d.up[i] = (float)w.m[4 + i]; d.direction[i] = (float)-w.m[8 + i]; }
        strncpy(d.name, c.name.c_str(), sizeof(d.name) - 1);
        return d;
    }
    // the records the lights become under `parent` (identity for a bare glTF file, the model node's transform in a .scene.json graph)
    static void emit_punctual(const PunctualRaw& l, const M4& parent, std::vector<PtAnalyticLightDesc>& analytic, std::vector<PtEnvDirectionalLight>& directional) {
        const M4 w = m4_mul(parent, l.world);
        const double zx = w.m[8], zy = w.m[9], zz = w.m[10]; double len = sqrt(zx * zx + zy * zy + zz * zz); if (!(len > 0)) len = 1;
        if (l.type == 2) {
            PtEnvDirectionalLight d; memset(&d, 0, sizeof(d));
            d.ColorIntensity[0] = l.color[0]; d.ColorIntensity[1] = l.color[1]; d.ColorIntensity[2] = l.color[2]; d.ColorIntensity[3] = l.intensity;
            d.Direction[0] = (float)(-zx / len); d.Direction[1] = (float)(-zy / len); d.Direction[2] = (float)(-zz / len); d.AngularSize = 0.f;
            directional.push_back(d);
        } else {
            PtAnalyticLightDesc d; memset(&d, 0, sizeof(d));
            d.type = (uint32_t)l.type; for (int i = 0; i < 3; i++) { d.color[i] = l.color[i]; d.position[i] = (float)w.m[12 + i]; }
            d.intensity = l.intensity; d.radius = 0.f; d.innerAngle = l.inner; d.outerAngle = l.outer;
            d.direction[0] = (float)(-zx / len); d.direction[1] = (float)(-zy / len); d.direction[2] = (float)(-zz / len);
            analytic.push_back(d);
        }
    }
    struct NodeTRS { bool has[3]; double t[3], q[4], s[3]; };      // animation: per node, the channels that replace its translation / rotation / scale (pt_gltf_animation)
    const std::vector<NodeTRS>* nodeOverride = nullptr;

    // `count` elements of `comps` components of type `ct` from bufferView `bv` (+ byteOffset acOff) into out; tight = ignore the view's byteStride (sparse indices / values are tightly packed)
    bool read_view(int bv, double acOff, int ct, int comps, int count, bool normalized, bool tight, std::vector<double>& out) {
        const JValue* bvs = root.get("bufferViews");
        if (!bvs || bv < 0 || (size_t)bv >= bvs->size()) { err = "accessor without bufferView"; return false; }
        const JValue& v = bvs->arr[bv];
        const double bvOff = v.numOr("byteOffset", 0), bvStride = v.numOr("byteStride", 0);
        if (count < 0 || !(bvOff >= 0) || !(acOff >= 0) || !(bvStride >= 0) || bvOff > 4.0e12 || acOff > 4.0e12 || bvStride > 65536.0) { err = "accessor with a negative or absurd count / offset / stride"; return false; }
        int buf = v.intOr("buffer", 0); size_t off = (size_t)bvOff + (size_t)acOff;
        int csz = (ct == 5120 || ct == 5121) ? 1 : (ct == 5122 || ct == 5123) ? 2 : (ct == 5125 || ct == 5126) ? 4 : 0;
        if (!csz || buf < 0 || (size_t)buf >= buffers.size()) { err = "unsupported component type"; return false; }
        size_t stride = tight ? 0 : (size_t)bvStride; if (!stride) stride = (size_t)csz * comps;
        const std::vector<uint8_t>& b = buffers[buf];
        const size_t elem = (size_t)csz * comps;
        if (count && (off > b.size() || elem > b.size() - off || (size_t)(count - 1) > (b.size() - off - elem) / stride)) { err = "accessor out of range"; return false; }
        out.resize((size_t)count * comps);
        for (int i = 0; i < count; i++) for (int k = 0; k < comps; k++) {
            const uint8_t* p = &b[off + stride * i + (size_t)csz * k]; double val;
            switch (ct) { case 5120: val = *(const int8_t*)p; if (normalized) val = val / 127.0 < -1 ? -1 : val / 127.0; break; case 5121: val = *p; if (normalized) val /= 255.0; break;
                case 5122: { int16_t s; memcpy(&s, p, 2); val = s; if (normalized) val = val / 32767.0 < -1 ? -1 : val / 32767.0; } break;
                case 5123: { uint16_t s; memcpy(&s, p, 2); val = s; if (normalized) val /= 65535.0; } break;
                case 5125: { uint32_t s; memcpy(&s, p, 4); val = s; } break; default: { float f; memcpy(&f, p, 4); val = f; } break; }
            out[(size_t)i * comps + k] = val;
        }
        return true;
    }
    bool accessor(int idx, std::vector<double>& out, int& comps) {
        const JValue* accs = root.get("accessors"); if (!accs || idx < 0 || (size_t)idx >= accs->size()) { err = "bad accessor index"; return false; }
        const JValue& a = accs->arr[idx];
        int ct = a.intOr("componentType", 0), count = a.intOr("count", 0); std::string type = a.strOr("type", "");
        comps = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
        bool normalized = a.get("normalized") && a.get("normalized")->b;
        if (!comps) { err = "unsupported accessor"; return false; }
        const JValue* sparse = a.get("sparse");
        if (count < 0 || count > (1 << 28)) { err = "accessor with a negative or absurd count / offset / stride"; return false; }
        if (a.get("bufferView")) { if (!read_view(a.intOr("bufferView", -1), a.numOr("byteOffset", 0), ct, comps, count, normalized, false, out)) return false; }
        else if (sparse) {      // glTF 2.0, 3.6.2.3: a sparse accessor without a bufferView starts from zeros
            // ... of which there is no buffer to bound the count by (a few bytes of JSON with count = 2^28 and MAT4 would ask for 32 GiB of zeros): no more elements than every vertex
            // of a 2^26-component stream, the bound the file's own buffers put on any other accessor of a scene this library can hold
            if ((size_t)count * (size_t)comps > ((size_t)1 << 26)) { err = "sparse accessor without a bufferView and with an absurd count"; return false; }
            out.assign((size_t)count * comps, 0.0);
        }
        else { err = "accessor without bufferView"; return false; }
        if (sparse) {      // sparse.count elements are replaced: sparse.indices (tightly packed u8 / u16 / u32) name them, sparse.values (tightly packed, the accessor's type) hold them.
            // cgltf — what Donut's importer reads glTF with — resolves sparse accessors when it unpacks floats (cgltf_accessor_unpack_floats), so the reference accepts such files
            const int n = sparse->intOr("count", 0); const JValue* ind = sparse->get("indices"); const JValue* val = sparse->get("values");
            if (n < 0 || n > count || !ind || !val) { err = "malformed sparse accessor"; return false; }
            const int ict = ind->intOr("componentType", 0);
            if (ict != 5121 && ict != 5123 && ict != 5125) { err = "malformed sparse accessor"; return false; }
            std::vector<double> idxs, vals;
            if (!read_view(ind->intOr("bufferView", -1), ind->numOr("byteOffset", 0), ict, 1, n, false, true, idxs)) return false;
            if (!read_view(val->intOr("bufferView", -1), val->numOr("byteOffset", 0), ct, comps, n, normalized, true, vals)) return false;
            for (int i = 0; i < n; i++) {
                const double at = idxs[(size_t)i];
                if (!(at >= 0) || at >= (double)count) { err = "sparse accessor index out of range"; return false; }
                for (int k = 0; k < comps; k++) out[(size_t)at * comps + k] = vals[(size_t)i * comps + k];
            }
        }
        return true;
    }
    uint32_t texture(const JValue* texRef, bool srgb) {          // returns packed texture word or 0xFFFFFFFF
        if (!texRef) return 0xFFFFFFFFu;
        int ti = texRef->intOr("index", -1); const JValue* texs = root.get("textures"); const JValue* imgs = root.get("images");
        if (!texs || !imgs || ti < 0 || (size_t)ti >= texs->size()) return 0xFFFFFFFFu;
        int img = texs->arr[ti].intOr("source", -1);
        if (const JValue* ext = texs->arr[ti].get("extensions")) if (const JValue* dds = ext->get("MSFT_texture_dds")) { int s = dds->intOr("source", -1); if (s >= 0) img = s; }      // Donut's importer prefers the .dds image (what Bistro's glTF carries)
        if (img < 0 || (size_t)img >= imgs->size()) return 0xFFFFFFFFu;
        auto key = std::make_pair(img, srgb ? 1 : 0); auto it = texCache.find(key); if (it != texCache.end()) return it->second;
        const JValue& im = imgs->arr[img]; std::vector<uint8_t> file;
        if (im.get("uri")) { if (!load_uri(baseDir, im.strOr("uri", ""), file)) return 0xFFFFFFFFu; }
        else {
            int bv = im.intOr("bufferView", -1); const JValue* bvs = root.get("bufferViews"); if (!bvs || bv < 0 || (size_t)bv >= bvs->size()) return 0xFFFFFFFFu;
            const JValue& v = bvs->arr[bv]; int buf = v.intOr("buffer", 0); const double dOff = v.numOr("byteOffset", 0), dLen = v.numOr("byteLength", 0);
            if (buf < 0 || (size_t)buf >= buffers.size() || !(dOff >= 0) || !(dLen >= 0) || dOff > 4e12 || dLen > 4e12) return 0xFFFFFFFFu;      // (NaN, negative and huge values never reach the casts)
            const size_t off = (size_t)dOff, len = (size_t)dLen, have = buffers[buf].size();
            if (off > have || len > have - off) return 0xFFFFFFFFu;                                                                    // no sum that could wrap
            file.assign(buffers[buf].begin() + off, buffers[buf].begin() + off + len);
        }
        uint32_t w, h; std::vector<uint8_t> rgba;
        if (file.size() > 4 && !memcmp(file.data(), "DDS ", 4)) {          // image/vnd-ms.dds (pt_dds.cpp); the sRGB-ness is the texture slot's, as for the other formats
            uint32_t fmt = 0; void* px = nullptr; if (pt_image_read_dds_memory(file.data(), file.size(), &w, &h, &fmt, &px) != PT_OK) return 0xFFFFFFFFu;
            if (fmt == PT_TEX_RGBA32F) { pt_image_free((float*)px); return 0xFFFFFFFFu; }
            { PxGuard guard{px}; rgba.assign((const uint8_t*)px, (const uint8_t*)px + (size_t)w * h * 4u); }
        }
        else if (file.size() > 2 && file[0] == 0xFF && file[1] == 0xD8) {      // image/jpeg (pt_jpeg.cpp)
            void* px = nullptr; if (pt_image_read_jpeg(file.data(), file.size(), &w, &h, &px) != PT_OK) return 0xFFFFFFFFu;
            { PxGuard guard{px}; rgba.assign((const uint8_t*)px, (const uint8_t*)px + (size_t)w * h * 4u); }
        }
        else if (!decode_png(file, w, h, rgba)) return 0xFFFFFFFFu;      // other image formats are treated as "texture not loaded"
        uint32_t index = (uint32_t)texDescs.size();
        texPixels.push_back(std::move(rgba));
        PtTextureDesc d; d.width = w; d.height = h; d.format = srgb ? PT_TEX_RGBA8_SRGB : PT_TEX_RGBA8_UNORM; d.pixels = nullptr; texDescs.push_back(d);
        uint32_t word = pack_texture_word(index, w, h); texCache[key] = word; return word;
    }
    // MaterialsBaker::ImportFromDonut (MaterialsBaker.cpp:660-705) over what Donut's glTF importer puts into donut::engine::Material, then
    // PTMaterial::FillData. The reference takes from the glTF document: the five textures, base colour / opacity, emissive colour and strength,
    // metalness, roughness, alpha cutoff, transmission factor, normal scale, and the domain (alpha tested / transmissive). It does NOT import the index
    // of refraction (`//materialPT->IoR = material.ior`), the diffuse transmission factor or any volume attenuation: those stay at the PTMaterial
    // defaults (1.5, 0, white / FLT_MAX), ThinSurface stays false — a KHR_materials_transmission material is a refracting solid of priority 14 —
    // and so they do here (KHR_materials_ior / KHR_materials_volume are read past).
    void importMaterials() {
        const JValue* mats = root.get("materials"); size_t n = mats ? mats->size() : 0;
        for (size_t i = 0; i <= n; i++) {            // one extra default material for primitives without one
            PTMaterialHost m; m.Roughness = 1.f; m.Metalness = 1.f;          // glTF pbrMetallicRoughness defaults (Donut fills its Material from the document)
            uint32_t words[5] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};      // base, metal-rough, normal, emissive, transmission
            if (i == n) { m.Metalness = 0.f; }
            else {
                const JValue& j = mats->arr[i];
                if (const JValue* pbr = j.get("pbrMetallicRoughness")) {
                    if (const JValue* c = pbr->get("baseColorFactor")) if (c->size() >= 3) { for (int k = 0; k < 3; k++) m.BaseOrDiffuseColor[k] = (float)c->arr[k].num; if (c->size() > 3) m.Opacity = (float)c->arr[3].num; }
                    m.Metalness = (float)pbr->numOr("metallicFactor", 1.0); m.Roughness = (float)pbr->numOr("roughnessFactor", 1.0);
                    words[0] = texture(pbr->get("baseColorTexture"), true); words[1] = texture(pbr->get("metallicRoughnessTexture"), false);
                }
                if (const JValue* e = j.get("emissiveFactor")) if (e->size() >= 3) for (int k = 0; k < 3; k++) m.EmissiveColor[k] = (float)e->arr[k].num;
                words[3] = texture(j.get("emissiveTexture"), true);
                if (const JValue* nt = j.get("normalTexture")) { words[2] = texture(nt, false); if (words[2] != 0xFFFFFFFFu) m.NormalTextureScale = (float)nt->numOr("scale", 1.0); }
                m.AlphaCutoff = (float)j.numOr("alphaCutoff", 0.5);
                m.EnableAlphaTesting = j.strOr("alphaMode", "OPAQUE") == "MASK";                 // MaterialDomain::AlphaTested
                if (const JValue* ext = j.get("extensions")) {
                    if (const JValue* es = ext->get("KHR_materials_emissive_strength")) m.EmissiveIntensity = (float)es->numOr("emissiveStrength", 1.0);
                    // KHR_materials_pbrSpecularGlossiness (what Bistro ships): Donut's importer prefers it over pbrMetallicRoughness and fills useSpecularGlossModel, the
                    // diffuse / specular colours, roughness = 1 - glossiness and BOTH textures as sRGB (ImportFromDonut, MaterialsBaker.cpp:669-670, 698)
                    if (const JValue* sg = ext->get("KHR_materials_pbrSpecularGlossiness")) {
                        m.UseSpecularGlossModel = true; m.Metalness = 0.f; m.Opacity = 1.f;
                        for (int k = 0; k < 3; k++) { m.BaseOrDiffuseColor[k] = 1.f; m.SpecularColor[k] = 1.f; }
                        if (const JValue* c = sg->get("diffuseFactor")) if (c->size() >= 3) { for (int k = 0; k < 3; k++) m.BaseOrDiffuseColor[k] = (float)c->arr[k].num; if (c->size() > 3) m.Opacity = (float)c->arr[3].num; }
                        if (const JValue* c = sg->get("specularFactor")) if (c->size() >= 3) for (int k = 0; k < 3; k++) m.SpecularColor[k] = (float)c->arr[k].num;
                        m.Roughness = 1.f - (float)sg->numOr("glossinessFactor", 1.0);
                        words[0] = texture(sg->get("diffuseTexture"), true); words[1] = texture(sg->get("specularGlossinessTexture"), true);
                    }
                    if (const JValue* tr = ext->get("KHR_materials_transmission")) {
                        m.TransmissionFactor = (float)tr->numOr("transmissionFactor", 0.0);
                        words[4] = texture(tr->get("transmissionTexture"), false);
                        m.EnableTransmission = m.TransmissionFactor > 0.f || words[4] != 0xFFFFFFFFu;      // MaterialDomain::Transmissive*
                    }
                }
            }
            bool loaded[5]; for (int t = 0; t < 5; t++) loaded[t] = words[t] != 0xFFFFFFFFu;
            PTMaterialData d; fill_data(m, loaded, words, &d);
            materials.push_back(d);
        }
    }
    bool importMeshes() {
        const JValue* ms = root.get("meshes"); size_t n = ms ? ms->size() : 0; const JValue* mats = root.get("materials"); size_t nMat = mats ? mats->size() : 0;
        meshMap.assign(n, -1);
        for (size_t mi = 0; mi < n; mi++) {
            const JValue* prims = ms->arr[mi].get("primitives"); if (!prims) continue;
            uint32_t firstGeom = (uint32_t)geoms.size();
            for (auto& pr : prims->arr) {
                if (pr.intOr("mode", 4) != 4) continue;
                const JValue* at = pr.get("attributes"); if (!at || !at->get("POSITION")) continue;
                std::vector<double> P, N, T, UV, I, J, Wt; int c;
                if (!accessor(at->intOr("POSITION", -1), P, c) || c != 3) { if (err.empty()) err = "POSITION must be VEC3"; return false; }
                uint32_t nv = (uint32_t)(P.size() / 3), flags = 0;
                if (at->get("NORMAL")) { if (!accessor(at->intOr("NORMAL", -1), N, c) || c != 3 || N.size() / 3 != nv) return false; flags |= PT_GEOM_HAS_NORMAL; }
                if (at->get("TANGENT")) { if (!accessor(at->intOr("TANGENT", -1), T, c) || c != 4 || T.size() / 4 != nv) return false; flags |= PT_GEOM_HAS_TANGENT; }
                if (at->get("TEXCOORD_0")) { if (!accessor(at->intOr("TEXCOORD_0", -1), UV, c) || c != 2 || UV.size() / 2 != nv) return false; flags |= PT_GEOM_HAS_UV; }
                const bool hasSkin = at->get("JOINTS_0") && at->get("WEIGHTS_0");
                if (hasSkin) { if (!accessor(at->intOr("JOINTS_0", -1), J, c) || c != 4 || J.size() / 4 != nv) return false; if (!accessor(at->intOr("WEIGHTS_0", -1), Wt, c) || c != 4 || Wt.size() / 4 != nv) return false; }
                if (pr.get("indices")) { if (!accessor(pr.intOr("indices", -1), I, c) || c != 1) return false; } else { I.resize(nv); for (uint32_t k = 0; k < nv; k++) I[k] = k; }
                uint32_t ni = (uint32_t)(I.size() / 3) * 3;
                GeomMorph gm{morphDeltas.size(), 0u};
                if (const JValue* tg = pr.get("targets")) for (auto& tj : tg->arr) {
                    std::vector<double> D;
                    if (tj.get("POSITION")) { if (!accessor(tj.intOr("POSITION", -1), D, c) || c != 3 || D.size() / 3 != nv) { if (err.empty()) err = "morph target POSITION must be VEC3, one per vertex"; return false; } }
                    else D.assign(3 * (size_t)nv, 0.0);                                  // a target without positions displaces nothing here (normals and tangents keep their base values)
                    for (double x : D) morphDeltas.push_back((float)x);
                    for (int which = 0; which < 2; which++) {      // NORMAL / TANGENT displacements (VEC3 each; absent: zeros), same layout as the positions'
                        const char* key = which ? "TANGENT" : "NORMAL"; std::vector<double> E;
                        if (tj.get(key)) { if (!accessor(tj.intOr(key, -1), E, c) || c != 3 || E.size() / 3 != nv) { if (err.empty()) err = "morph target NORMAL / TANGENT must be VEC3, one per vertex"; return false; } }
                        else E.assign(3 * (size_t)nv, 0.0);
                        std::vector<float>& dst = which ? morphTangentDeltas : morphNormalDeltas; for (double x : E) dst.push_back((float)x);
                    }
                    gm.count++;
                }
                geomMorph.push_back(gm);
                PtGeometryDesc g; memset(&g, 0, sizeof(g));
                g.indexOffset = (uint32_t)indices.size(); g.numIndices = ni; g.vertexOffset = (uint32_t)(positions.size() / 3); g.numVertices = nv; g.flags = flags;
                int mat = pr.intOr("material", -1); g.materialIndex = (mat >= 0 && (size_t)mat < nMat) ? (uint32_t)mat : (uint32_t)nMat;
                if (mat >= 0 && (size_t)mat < nMat && mats->arr[mat].strOr("alphaMode", "OPAQUE") == "MASK") g.geomFlags |= PT_GEOMF_ALPHA_TESTED;   // MaterialDomain::AlphaTested
                for (uint32_t k = 0; k < ni; k++) { uint32_t v = (uint32_t)I[k]; if (v >= nv) { err = "index out of range"; return false; } indices.push_back(v); }
                for (uint32_t k = 0; k < nv; k++) {
                    positions.push_back((float)P[3 * k]); positions.push_back((float)P[3 * k + 1]); positions.push_back((float)P[3 * k + 2]);
                    float uv[2] = {0, 0}; if (flags & PT_GEOM_HAS_UV) { uv[0] = (float)UV[2 * k]; uv[1] = (float)UV[2 * k + 1]; } uvs.push_back(uv[0]); uvs.push_back(uv[1]);
                    float nn[3] = {0, 0, 0}; if (flags & PT_GEOM_HAS_NORMAL) { nn[0] = (float)N[3 * k]; nn[1] = (float)N[3 * k + 1]; nn[2] = (float)N[3 * k + 2]; } normals.push_back(pack_snorm8(nn, 3)); normalsF.insert(normalsF.end(), nn, nn + 3);
                    float tt[4] = {0, 0, 0, 0}; if (flags & PT_GEOM_HAS_TANGENT) { for (int q = 0; q < 4; q++) tt[q] = (float)T[4 * k + q]; } tangents.push_back(pack_snorm8(tt, 4)); tangentsF.insert(tangentsF.end(), tt, tt + 4);
                    for (int q = 0; q < 4; q++) { const double jv = hasSkin ? J[4 * k + q] : 0.0; joints.push_back((uint16_t)(jv >= 0 && jv <= 65535.0 ? jv : 0)); weights.push_back(hasSkin ? (float)Wt[4 * k + q] : 0.f); }
                }
                geoms.push_back(g);
            }
            if (geoms.size() > firstGeom) { PtMeshDesc md; md.firstGeometry = firstGeom; md.numGeometries = (uint32_t)geoms.size() - firstGeom; meshMap[mi] = (int)meshes.size(); meshes.push_back(md);
                std::vector<double> w; if (const JValue* jw = ms->arr[mi].get("weights")) for (auto& x : jw->arr) w.push_back(x.num); meshWeights.push_back(std::move(w)); }
        }
        return true;
    }
    void visit(int node, const M4& parent, int depth, const std::string& parentPath = std::string()) {
        const JValue* nodes = root.get("nodes"); if (!nodes || node < 0 || (size_t)node >= nodes->size() || depth > 256) return;
        const JValue& n = nodes->arr[node]; M4 local = m4_identity();
        const std::string path = parentPath.empty() ? n.strOr("name", "") : parentPath + "/" + n.strOr("name", "");
        const NodeTRS* ov = (nodeOverride && (size_t)node < nodeOverride->size()) ? &(*nodeOverride)[(size_t)node] : nullptr;
        const bool animated = ov && (ov->has[0] || ov->has[1] || ov->has[2]);
        if (const JValue* mx = n.get("matrix"); mx && !animated) { if (mx->size() == 16) for (int i = 0; i < 16; i++) local.m[i] = mx->arr[i].num; }      // (glTF: an animated node has TRS properties, not a matrix)
        else {
            double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
            if (const JValue* v = n.get("translation")) if (v->size() == 3) for (int i = 0; i < 3; i++) t[i] = v->arr[i].num;
            if (const JValue* v = n.get("rotation")) if (v->size() == 4) for (int i = 0; i < 4; i++) q[i] = v->arr[i].num;
            if (const JValue* v = n.get("scale")) if (v->size() == 3) for (int i = 0; i < 3; i++) s[i] = v->arr[i].num;
            if (ov) { if (ov->has[0]) memcpy(t, ov->t, sizeof(t)); if (ov->has[1]) memcpy(q, ov->q, sizeof(q)); if (ov->has[2]) memcpy(s, ov->s, sizeof(s)); }
            local = m4_trs(t, q, s);
        }
        M4 world = m4_mul(parent, local);
        int mesh = n.intOr("mesh", -1);
        if (recordWorlds) { if (nodeWorld.size() < nodes->size()) nodeWorld.resize(nodes->size(), m4_identity()); nodeWorld[(size_t)node] = world;
                            if (mesh >= 0 && (size_t)mesh < meshMap.size() && meshMap[mesh] >= 0) meshNodes.push_back({node, meshMap[mesh]});
                            if (mesh >= 0 && (size_t)mesh < meshMap.size() && meshMap[mesh] >= 0 && n.intOr("skin", -1) >= 0) skinned.push_back({node, n.intOr("skin", -1), meshMap[mesh]}); }
        if (mesh >= 0 && (size_t)mesh < meshMap.size() && meshMap[mesh] >= 0) {
            PtInstanceDesc inst; memset(&inst, 0, sizeof(inst)); inst.meshIndex = (uint32_t)meshMap[mesh];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) inst.transform[r * 4 + c] = (float)world.m[c * 4 + r];   // column-major 4x4 -> row-major 3x4
            instances.push_back(inst); instanceWorld.push_back(world); instancePath.push_back(path);
        }
        if (const JValue* ext = n.get("extensions")) if (const JValue* lp = ext->get("KHR_lights_punctual")) punctual_light(lp->intOr("light", -1), world);
        if (n.get("camera")) gltf_camera(n.intOr("camera", -1), world, n.strOr("name", ""));
        if (const JValue* ch = n.get("children")) for (auto& c : ch->arr) visit((int)c.num, world, depth + 1, path);
    }
};

void jload(const JValue& o, const char* k, float& v) { const JValue* j = o.get(k); if (j && j->type == JValue::Num) v = (float)j->num; }
void jload(const JValue& o, const char* k, int& v) { const JValue* j = o.get(k); if (j && j->type == JValue::Num) v = (int)j->num; }
void jload(const JValue& o, const char* k, bool& v) { const JValue* j = o.get(k); if (j && j->type == JValue::Bool) v = j->b; else if (j && j->type == JValue::Num) v = j->num != 0; }
void jload3(const JValue& o, const char* k, float v[3]) { const JValue* j = o.get(k); if (j && j->type == JValue::Arr && j->arr.size() >= 3) for (int i = 0; i < 3; i++) if (j->arr[i].type == JValue::Num) v[i] = (float)j->arr[i].num; }

} // namespace

static int32_t material_from_json_impl(const char* jsonText, const uint32_t textureWords[5], PTMaterialData* out, PtMaterialJsonInfo* info);
extern "C" int32_t pt_material_from_json(const char* jsonText, const uint32_t textureWords[5], PTMaterialData* out, PtMaterialJsonInfo* info) {
    try { return material_from_json_impl(jsonText, textureWords, out, info); } catch (...) { return PT_ERROR_IO; }      // no exception crosses the C ABI
}
static int32_t material_from_json_impl(const char* jsonText, const uint32_t textureWords[5], PTMaterialData* out, PtMaterialJsonInfo* info) {
    if (!jsonText || !out) return PT_ERROR_INVALID_ARGUMENT;
    JParser jp{jsonText, jsonText + strlen(jsonText)};
    JValue root = jp.parse();
    if (!jp.ok || root.type != JValue::Obj) return PT_ERROR_IO;
    PTMaterialHost m;
#define PT_LOAD_FIELD(NAME) jload(root, #NAME, m.NAME)
    jload3(root, "BaseOrDiffuseColor", m.BaseOrDiffuseColor); jload3(root, "SpecularColor", m.SpecularColor); jload3(root, "EmissiveColor", m.EmissiveColor);
    PT_LOAD_FIELD(EmissiveIntensity); PT_LOAD_FIELD(Metalness); PT_LOAD_FIELD(Roughness); PT_LOAD_FIELD(Opacity); PT_LOAD_FIELD(TransmissionFactor); PT_LOAD_FIELD(DiffuseTransmissionFactor);
    PT_LOAD_FIELD(NormalTextureScale); PT_LOAD_FIELD(IoR); PT_LOAD_FIELD(UseSpecularGlossModel); PT_LOAD_FIELD(EnableBaseTexture); PT_LOAD_FIELD(EnableOcclusionRoughnessMetallicTexture);
    PT_LOAD_FIELD(EnableNormalTexture); PT_LOAD_FIELD(EnableEmissiveTexture); PT_LOAD_FIELD(EnableTransmissionTexture); PT_LOAD_FIELD(EnableAlphaTesting); PT_LOAD_FIELD(AlphaCutoff);
    PT_LOAD_FIELD(EnableTransmission); PT_LOAD_FIELD(MetalnessInRedChannel); PT_LOAD_FIELD(ThinSurface); PT_LOAD_FIELD(ExcludeFromNEE); PT_LOAD_FIELD(PSDExclude);
    PT_LOAD_FIELD(PSDBlockMotionVectorsAtSurfaceType); PT_LOAD_FIELD(PSDDominantDeltaLobe); PT_LOAD_FIELD(NestedPriority); PT_LOAD_FIELD(VolumeAttenuationDistance);
    jload3(root, "VolumeAttenuationColor", m.VolumeAttenuationColor); PT_LOAD_FIELD(ShadowNoLFadeout); PT_LOAD_FIELD(EnableAsAnalyticLightProxy); PT_LOAD_FIELD(IgnoreMeshTangentSpace);
    PT_LOAD_FIELD(UseDonutEmissiveIntensity); PT_LOAD_FIELD(SkipRender);
#undef PT_LOAD_FIELD
    static const char* texNames[5] = {"BaseTexture", "OcclusionRoughnessMetallicTexture", "NormalTexture", "EmissiveTexture", "TransmissionTexture"};
    bool loaded[5];
    if (info) memset(info, 0, sizeof(*info));
    for (int t = 0; t < 5; t++) {
        const JValue* tj = root.get(texNames[t]);
        std::string path = (tj && tj->type == JValue::Obj) ? tj->strOr("path", "") : std::string();
        // "Loaded" in the reference = the texture object exists; here: the document names a path AND the caller supplied a texture word for it
        loaded[t] = !path.empty() && textureWords && textureWords[t] != 0xFFFFFFFFu;
        if (info && tj && tj->type == JValue::Obj) {
            strncpy(info->texturePath[t], path.c_str(), 255);
            bool b = false; jload(*tj, "sRGB", b); info->textureSRGB[t] = b; b = false; jload(*tj, "NormalMap", b); info->textureNormalMap[t] = b;
        }
    }
    fill_data(m, loaded, textureWords, out);      // FillData (MaterialsBaker.cpp:516-591)
    if (info) { info->enableAlphaTesting = m.EnableAlphaTesting; info->excludeFromNEE = m.ExcludeFromNEE; info->skipRender = m.SkipRender; info->useDonutEmissiveIntensity = m.UseDonutEmissiveIntensity; }
    return PT_OK;
}

namespace {
} // namespace

namespace {
// one glTF / GLB file -> Loader arrays (materials, textures, geometry, meshes, instances of the default scene)
int32_t load_gltf_file(const char* path, Loader& L) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return PT_ERROR_IO;
    std::string sp(path); size_t slash = sp.find_last_of('/'); L.baseDir = slash == std::string::npos ? "" : sp.substr(0, slash + 1);
    const char* jb = (const char*)file.data(); size_t jn = file.size(); std::vector<uint8_t> glbBin;
    if (file.size() >= 20 && !memcmp(file.data(), "glTF", 4)) {                       // GLB container
        uint32_t jlen; memcpy(&jlen, &file[12], 4); if (20 + (size_t)jlen > file.size()) return PT_ERROR_IO;
        jb = (const char*)&file[20]; jn = jlen; size_t p = 20 + jlen;
        if (p + 8 <= file.size()) { uint32_t blen; memcpy(&blen, &file[p], 4); if (p + 8 + blen <= file.size()) glbBin.assign(file.begin() + p + 8, file.begin() + p + 8 + blen); }
    }
    JParser jp; jp.p = jb; jp.e = jb + jn; L.root = jp.parse();
    if (!jp.ok || L.root.type != JValue::Obj) return PT_ERROR_IO;
    if (const JValue* bufs = L.root.get("buffers")) for (auto& b : bufs->arr) {
        std::vector<uint8_t> data;
        if (b.get("uri")) { if (!load_uri(L.baseDir, b.strOr("uri", ""), data)) return PT_ERROR_IO; } else data = glbBin;
        L.buffers.push_back(std::move(data));
    }
    L.importMaterials();
    if (!L.importMeshes()) return PT_ERROR_IO;
    int sceneIdx = L.root.intOr("scene", 0); const JValue* scenes = L.root.get("scenes");
    if (scenes && (size_t)sceneIdx < scenes->size()) { if (const JValue* ns = scenes->arr[sceneIdx].get("nodes")) for (auto& n : ns->arr) L.visit((int)n.num, m4_identity(), 0); }
    else for (int r : gltf_parentless_nodes(L.root.get("nodes"))) L.visit(r, m4_identity(), 0);
    return PT_OK;
}
} // namespace

static int32_t load_scene_gltf_impl(pt_context* ctx, const char* path);
extern "C" int32_t pt_load_scene_gltf(pt_context* ctx, const char* path) {
    try { return load_scene_gltf_impl(ctx, path); } catch (...) { return PT_ERROR_IO; }      // bad_alloc / length_error on a damaged file: an error code, not an abort
}
static int32_t load_scene_json_into(pt_context* ctx, const char* path);
static int32_t load_scene_gltf_impl(pt_context* ctx, const char* path) {
    if (!ctx || !path) return PT_ERROR_INVALID_ARGUMENT;
    { const size_t n = strlen(path); if (n >= 5 && strcmp(path + n - 5, ".json") == 0) return load_scene_json_into(ctx, path); }      // an RTXPT asset folder's `.scene.json`
    Loader L; L.ctx = ctx;
    int32_t lr = load_gltf_file(path, L);
    if (lr != PT_OK) return lr;
    if (L.instances.empty() || L.geoms.empty()) return PT_ERROR_IO;
    for (size_t i = 0; i < L.texDescs.size(); i++) L.texDescs[i].pixels = L.texPixels[i].data();
    int32_t r = pt_set_materials(ctx, L.materials.data(), (uint32_t)L.materials.size(), L.texDescs.data(), (uint32_t)L.texDescs.size());
    if (r != PT_OK) return r;
    PtGeometryBuffers gb; gb.indices = L.indices.data(); gb.numIndices = (uint32_t)L.indices.size(); gb.positions = L.positions.data(); gb.uvs = L.uvs.data();
    gb.normals = L.normals.data(); gb.tangents = L.tangents.data(); gb.numVertices = (uint32_t)(L.positions.size() / 3);
    r = pt_set_geometry(ctx, &gb, L.geoms.data(), (uint32_t)L.geoms.size(), L.meshes.data(), (uint32_t)L.meshes.size());
    if (r != PT_OK) return r;
    r = pt_set_instances(ctx, L.instances.data(), (uint32_t)L.instances.size());
    if (r != PT_OK) return r;
    // KHR_lights_punctual: point / spot lights become the context's analytic lights (what LightsBaker collects from the scene graph), directional ones the scene's list for the
    // environment bake (pt_set_scene_directional_lights: Sample::UpdateLighting's conversion runs at bake time). A file without the extension clears both.
    std::vector<PtAnalyticLightDesc> analytic; std::vector<PtEnvDirectionalLight> directional;
    for (const Loader::PunctualRaw& l : L.punctual) Loader::emit_punctual(l, m4_identity(), analytic, directional);
    // A scene load REPLACES the scene's lights, also with nothing (Sample::SceneLoaded rebuilds m_lights on every load, Sample.cpp:553-575): a second file loaded into the same
    // context must not inherit the first one's analytic lights or sun discs.
    {
        std::vector<PolymorphicLightInfo> base(analytic.size()); std::vector<PolymorphicLightInfoEx> ex(analytic.size());
        for (size_t i = 0; i < analytic.size(); i++) { r = pt_convert_light(&analytic[i], &base[i], &ex[i]); if (r != PT_OK) return r; }
        r = pt_set_lights(ctx, analytic.empty() ? nullptr : base.data(), analytic.empty() ? nullptr : ex.data(), (uint32_t)base.size()); if (r != PT_OK) return r;
    }
    r = pt_set_scene_directional_lights(ctx, directional.empty() ? nullptr : directional.data(), (uint32_t)std::min<size_t>(directional.size(), 16)); if (r != PT_OK) return r;
    return PT_OK;
}

// ================================================================ glTF animations (SURVEY.md 8f N2 leftovers)
// The reference animates through Donut's scene graph (Sample::Animate, Rtxpt/Sample.cpp:785-811 -> Scene::Animate, SceneGraphAnimation: not vendored); what a glTF file
// says about its animations is defined by the glTF 2.0 specification, which is what is implemented here: samplers with LINEAR (spherical for rotations), STEP and
// CUBICSPLINE interpolation over node translation / rotation / scale / weights channels, time clamped to the sampler's key range. Skins and morph targets: pt_gltf_animation_positions.
struct pt_gltf_animation {
    Loader L; bool parsed = false;
    struct Sampler { std::vector<double> in, out; int comps = 0; int mode = 0; };      // mode 0 LINEAR, 1 STEP, 2 CUBICSPLINE
    struct Channel { int sampler, node, path; };                                       // path 0 translation, 1 rotation, 2 scale, 3 weights (morph targets)
    struct Anim { std::vector<Sampler> samplers; std::vector<Channel> channels; double duration = 0; };
    std::vector<Anim> anims;
};
namespace {
void quat_normalize(double q[4]) { double l = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); if (l > 0) for (int i = 0; i < 4; i++) q[i] /= l; }
void sample_channel(const pt_gltf_animation::Sampler& sp, int path, double t, double* v) {
    const int n = path == 1 ? 4 : 3; const size_t keys = sp.in.size();
    const size_t stride = sp.mode == 2 ? (size_t)n * 3 : (size_t)n, valueAt = sp.mode == 2 ? (size_t)n : 0;      // CUBICSPLINE stores in-tangent, value, out-tangent per key
    auto value = [&](size_t k, double* o) { for (int i = 0; i < n; i++) o[i] = sp.out[k * stride + valueAt + (size_t)i]; };
    if (keys == 1 || t <= sp.in[0]) { value(0, v); if (path == 1) quat_normalize(v); return; }
    if (t >= sp.in[keys - 1]) { value(keys - 1, v); if (path == 1) quat_normalize(v); return; }
    size_t k = 0; while (k + 2 < keys && sp.in[k + 1] <= t) k++;
    const double t0 = sp.in[k], t1 = sp.in[k + 1], dt = t1 - t0, u = dt > 0 ? (t - t0) / dt : 0.0;
    double a[4], b[4]; value(k, a); value(k + 1, b);
    if (sp.mode == 1) { memcpy(v, a, sizeof(double) * (size_t)n); }
    else if (sp.mode == 2) {
        const double u2 = u * u, u3 = u2 * u;
        for (int i = 0; i < n; i++) { const double m0 = sp.out[k * stride + (size_t)n * 2 + (size_t)i] * dt, m1 = sp.out[(k + 1) * stride + (size_t)i] * dt;      // out-tangent of key k, in-tangent of key k + 1
            v[i] = (2 * u3 - 3 * u2 + 1) * a[i] + (u3 - 2 * u2 + u) * m0 + (-2 * u3 + 3 * u2) * b[i] + (u3 - u2) * m1; }
    } else if (path == 1) {                                                              // spherical linear interpolation along the shorter arc
        double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        if (d < 0) { d = -d; for (int i = 0; i < 4; i++) b[i] = -b[i]; }
        if (d > 0.9995) for (int i = 0; i < 4; i++) v[i] = a[i] + u * (b[i] - a[i]);
        else { const double th = acos(d), sn = sin(th), wa = sin((1 - u) * th) / sn, wb = sin(u * th) / sn; for (int i = 0; i < 4; i++) v[i] = wa * a[i] + wb * b[i]; }
    } else for (int i = 0; i < n; i++) v[i] = a[i] + u * (b[i] - a[i]);
    if (path == 1) quat_normalize(v);
}
// a "weights" channel: n scalars per key (n = the number of morph targets of the node's mesh), same three interpolation modes
bool sample_weights(const pt_gltf_animation::Sampler& sp, size_t n, double t, std::vector<double>& v) {
    const size_t keys = sp.in.size(), stride = sp.mode == 2 ? n * 3 : n, valueAt = sp.mode == 2 ? n : 0;
    if (!n || !keys || sp.out.size() != keys * stride) return false;
    v.assign(n, 0.0);
    auto value = [&](size_t k, size_t i) { return sp.out[k * stride + valueAt + i]; };
    if (keys == 1 || t <= sp.in[0]) { for (size_t i = 0; i < n; i++) v[i] = value(0, i); return true; }
    if (t >= sp.in[keys - 1]) { for (size_t i = 0; i < n; i++) v[i] = value(keys - 1, i); return true; }
    size_t k = 0; while (k + 2 < keys && sp.in[k + 1] <= t) k++;
    const double t0 = sp.in[k], t1 = sp.in[k + 1], dt = t1 - t0, u = dt > 0 ? (t - t0) / dt : 0.0, u2 = u * u, u3 = u2 * u;
    for (size_t i = 0; i < n; i++) {
        const double a = value(k, i), b = value(k + 1, i);
        if (sp.mode == 1) v[i] = a;
        else if (sp.mode == 2) { const double m0 = sp.out[k * stride + n * 2 + i] * dt, m1 = sp.out[(k + 1) * stride + i] * dt; v[i] = (2 * u3 - 3 * u2 + 1) * a + (u3 - 2 * u2 + u) * m0 + (-2 * u3 + 3 * u2) * b + (u3 - u2) * m1; }
        else v[i] = a + u * (b - a);
    }
    return true;
}
int32_t gltf_animation_load_impl(const char* path, pt_gltf_animation** out, uint32_t* numAnimations, float* duration) {
    if (!path || !out) return PT_ERROR_INVALID_ARGUMENT;
    *out = nullptr; if (numAnimations) *numAnimations = 0; if (duration) *duration = 0.f;
    std::unique_ptr<pt_gltf_animation> A(new pt_gltf_animation()); A->L.ctx = nullptr;
    int32_t r = load_gltf_file(path, A->L); if (r != PT_OK) return r;
    if (const JValue* anims = A->L.root.get("animations")) for (auto& ja : anims->arr) {
        pt_gltf_animation::Anim an;
        if (const JValue* ss = ja.get("samplers")) for (auto& js : ss->arr) {
            pt_gltf_animation::Sampler sp; int c = 0;
            if (!A->L.accessor(js.intOr("input", -1), sp.in, c) || c != 1 || sp.in.empty()) return PT_ERROR_IO;
            if (!A->L.accessor(js.intOr("output", -1), sp.out, sp.comps)) return PT_ERROR_IO;
            for (size_t k = 1; k < sp.in.size(); k++) if (!(sp.in[k] >= sp.in[k - 1])) return PT_ERROR_IO;      // key times must not decrease
            std::string m = js.strOr("interpolation", "LINEAR"); sp.mode = m == "STEP" ? 1 : m == "CUBICSPLINE" ? 2 : 0;
            an.samplers.push_back(std::move(sp));
        }
        if (const JValue* cs = ja.get("channels")) for (auto& jc : cs->arr) {
            const JValue* tg = jc.get("target"); if (!tg) continue;
            std::string pth = tg->strOr("path", ""); int pi = pth == "translation" ? 0 : pth == "rotation" ? 1 : pth == "scale" ? 2 : pth == "weights" ? 3 : -1;
            int node = tg->intOr("node", -1), smp = jc.intOr("sampler", -1);
            if (pi < 0 || node < 0) continue;                                            // (an unknown path, or a channel without a node: ignored as the specification allows)
            if (smp < 0 || (size_t)smp >= an.samplers.size()) return PT_ERROR_IO;
            if (pi == 3) {                                                               // scalars, (number of targets) per key: checked against the mesh when the weights are evaluated
                const pt_gltf_animation::Sampler& spw = an.samplers[(size_t)smp]; if (spw.comps != 1) return PT_ERROR_IO;
                an.channels.push_back({smp, node, pi}); if (spw.in.back() > an.duration) an.duration = spw.in.back(); continue;
            }
            const pt_gltf_animation::Sampler& sp = an.samplers[(size_t)smp]; const size_t n = pi == 1 ? 4 : 3;
            if ((size_t)sp.comps != n || sp.out.size() != sp.in.size() * n * (sp.mode == 2 ? 3u : 1u)) return PT_ERROR_IO;
            an.channels.push_back({smp, node, pi}); if (sp.in.back() > an.duration) an.duration = sp.in.back();
        }
        A->anims.push_back(std::move(an));
    }
    if (numAnimations) *numAnimations = (uint32_t)A->anims.size();
    if (duration) for (auto& an : A->anims) if ((float)an.duration > *duration) *duration = (float)an.duration;
    *out = A.release();
    return PT_OK;
}
} // namespace
extern "C" int32_t pt_gltf_animation_load(const char* path, pt_gltf_animation** out, uint32_t* numAnimations, float* duration) {
    try { return gltf_animation_load_impl(path, out, numAnimations, duration); } catch (...) { if (out) *out = nullptr; return PT_ERROR_IO; }
}
extern "C" void pt_gltf_animation_free(pt_gltf_animation* a) { delete a; }
// Skinned meshes (glTF 2.0 skins; Donut's SkinnedMeshInstance in the reference: Sample.cpp:1065, 1170-1198 rewrites the instance's vertex buffer every frame and updates its
// BLAS): the posed object-space positions of the WHOLE vertex stream — the `positions` argument of pt_animate. A vertex of a skinned primitive becomes
// SUM_k w_k (inverse(meshNodeWorld) * jointWorld_k * inverseBind_k) p, the joint matrices of the specification with the mesh node's own transform taken out (the instance keeps it).
// Morph targets (primitive.targets, POSITION displacements) are applied before the skin: p = base + SUM_i w_i target_i with the weights of the animation's "weights"
// channel, else of the node, else of the mesh. Normals and tangents: pt_gltf_animation_normals -> pt_animate_normals. A mesh shared by several nodes takes the pose of the last one.
// outN / outT (pt_gltf_animation_normals): the posed NORMAL / TANGENT streams in the packing pt_set_geometry takes (SNORM8 x 3 / x 4). A skinned vertex's normal is
// normalize(SUM_k w_k inverse-transpose(J_k) n), its tangent normalize(SUM_k w_k J_k t.xyz) with the handedness kept (glTF 2.0 3.7.3.3 leaves the normal transform to the
// implementation; the inverse transpose is the one that keeps normals perpendicular under the non-uniform scales a mesh node may carry). Morph targets' NORMAL / TANGENT displacements are added
// before the skin and the sums renormalised.
static int32_t gltf_animation_pose(pt_gltf_animation* a, uint32_t animation, float t, float* out, uint32_t* outN, uint32_t* outT, uint32_t capacityVertices) {
    if (!a || (capacityVertices && !out && !outN && !outT)) return -PT_ERROR_INVALID_ARGUMENT;
    try {
        Loader& L = a->L; const uint32_t nv = (uint32_t)(L.positions.size() / 3);
        if (capacityVertices < nv || nv == 0) return (int32_t)nv;                          // (query: the number of vertices; a file without vertices has nothing to pose)
        std::vector<float> posScratch; if (!out) { posScratch.resize(L.positions.size()); out = posScratch.data(); }
        memcpy(out, L.positions.data(), sizeof(float) * L.positions.size());
        if (outN) memcpy(outN, L.normals.data(), sizeof(uint32_t) * nv);
        if (outT) memcpy(outT, L.tangents.data(), sizeof(uint32_t) * nv);
        std::vector<float> nF, tF; const bool wantNT = outN || outT;      // the morphed (not yet skinned) normals / tangents, where a target displaces them
        if (wantNT) { nF = L.normalsF; tF = L.tangentsF; }
        const JValue* skins = L.root.get("skins"); const bool haveSkins = skins && skins->size();
        if (!haveSkins && L.morphDeltas.empty()) return (int32_t)nv;
        L.recordWorlds = true; L.skinned.clear(); L.nodeWorld.clear(); L.meshNodes.clear();
        std::vector<PtInstanceDesc> scratch(1); int32_t r = pt_gltf_animation_instances(a, animation, t, scratch.data(), 0);      // evaluates the channels and walks the nodes
        L.recordWorlds = false; if (r < 0) return r;
        // morph targets first (the specification applies them before the skin): p = base + SUM_i w_i * target_i. Weights: the animation's "weights" channel of the node,
        // else the node's own `weights`, else the mesh's
        if (!L.morphDeltas.empty()) {
            const JValue* nodes = L.root.get("nodes");
            for (const Loader::MeshNode& mn : L.meshNodes) {
                if ((size_t)mn.mesh >= L.meshes.size()) continue;
                const PtMeshDesc& md = L.meshes[(size_t)mn.mesh];
                uint32_t nT = 0; for (uint32_t g = 0; g < md.numGeometries; g++) if (L.geomMorph[md.firstGeometry + g].count > nT) nT = L.geomMorph[md.firstGeometry + g].count;
                if (!nT) continue;
                std::vector<double> w;
                if (animation < a->anims.size()) for (auto& c : a->anims[animation].channels) if (c.path == 3 && c.node == mn.node) sample_weights(a->anims[animation].samplers[(size_t)c.sampler], nT, (double)t, w);
                if (w.empty() && nodes && (size_t)mn.node < nodes->size()) if (const JValue* jw = nodes->arr[(size_t)mn.node].get("weights")) for (auto& x : jw->arr) w.push_back(x.num);
                if (w.empty()) w = L.meshWeights[(size_t)mn.mesh];
                w.resize(nT, 0.0);
                for (uint32_t g = 0; g < md.numGeometries; g++) {
                    const PtGeometryDesc& gd = L.geoms[md.firstGeometry + g]; const Loader::GeomMorph& gm = L.geomMorph[md.firstGeometry + g];
                    for (uint32_t v = 0; v < gd.numVertices; v++) for (int rr = 0; rr < 3; rr++) {
                        double p = L.positions[3 * (size_t)(gd.vertexOffset + v) + rr];
                        for (uint32_t i = 0; i < gm.count; i++) p += w[i] * (double)L.morphDeltas[gm.first + ((size_t)i * gd.numVertices + v) * 3 + rr];
                        out[3 * (size_t)(gd.vertexOffset + v) + rr] = (float)p;
                    }
                    if (wantNT) for (uint32_t v = 0; v < gd.numVertices; v++) {      // n = normalize(base + SUM_i w_i dn_i), likewise the tangent's xyz (glTF 2.0 3.7.2.2); repacked for the unskinned case
                        const size_t gv = gd.vertexOffset + v; double dn[3] = {0, 0, 0}, dt[3] = {0, 0, 0}; bool anyN = false, anyT = false;
                        for (uint32_t i = 0; i < gm.count; i++) for (int rr = 0; rr < 3; rr++) {
                            const float a = L.morphNormalDeltas[gm.first + ((size_t)i * gd.numVertices + v) * 3 + rr], b = L.morphTangentDeltas[gm.first + ((size_t)i * gd.numVertices + v) * 3 + rr];
                            dn[rr] += w[i] * (double)a; dt[rr] += w[i] * (double)b; anyN = anyN || (a != 0.f && w[i] != 0.0); anyT = anyT || (b != 0.f && w[i] != 0.0); }
                        if (anyN && (gd.flags & PT_GEOM_HAS_NORMAL)) { double q[3], l = 0; for (int rr = 0; rr < 3; rr++) { q[rr] = L.normalsF[3 * gv + rr] + dn[rr]; l += q[rr] * q[rr]; } l = sqrt(l);
                            if (l > 0) { for (int rr = 0; rr < 3; rr++) nF[3 * gv + rr] = (float)(q[rr] / l); if (outN) outN[gv] = pack_snorm8(&nF[3 * gv], 3); } }
                        if (anyT && (gd.flags & PT_GEOM_HAS_TANGENT)) { double q[3], l = 0; for (int rr = 0; rr < 3; rr++) { q[rr] = L.tangentsF[4 * gv + rr] + dt[rr]; l += q[rr] * q[rr]; } l = sqrt(l);
                            if (l > 0) { for (int rr = 0; rr < 3; rr++) tF[4 * gv + rr] = (float)(q[rr] / l); if (outT) outT[gv] = pack_snorm8(&tF[4 * gv], 4); } }
                    }
                }
            }
        }
        if (haveSkins) for (const Loader::SkinnedInstance& si : L.skinned) {
            if ((size_t)si.skin >= skins->size() || (size_t)si.mesh >= L.meshes.size()) continue;
            const JValue& sk = skins->arr[(size_t)si.skin]; const JValue* jl = sk.get("joints"); if (!jl || !jl->size()) continue;
            std::vector<double> ibm; int comps = 0; const bool haveIbm = sk.get("inverseBindMatrices") && L.accessor(sk.intOr("inverseBindMatrices", -1), ibm, comps) && comps == 16 && ibm.size() == 16 * jl->size();
            M4 invMesh; if (!m4_inverse(L.nodeWorld[(size_t)si.node], invMesh)) continue;
            std::vector<M4> jm(jl->size());
            for (size_t k = 0; k < jl->size(); k++) {
                const int jn = (int)jl->arr[k].num; M4 w = (jn >= 0 && (size_t)jn < L.nodeWorld.size()) ? L.nodeWorld[(size_t)jn] : m4_identity();
                M4 b = m4_identity(); if (haveIbm) memcpy(b.m, &ibm[16 * k], sizeof(b.m));
                jm[k] = m4_mul(invMesh, m4_mul(w, b));
            }
            const PtMeshDesc& md = L.meshes[(size_t)si.mesh];
            for (uint32_t g = 0; g < md.numGeometries; g++) {
                const PtGeometryDesc& gd = L.geoms[md.firstGeometry + g];
                for (uint32_t v = gd.vertexOffset; v < gd.vertexOffset + gd.numVertices; v++) {
                    const float* wgt = &L.weights[4 * (size_t)v]; const uint16_t* jnt = &L.joints[4 * (size_t)v];
                    if (!(wgt[0] + wgt[1] + wgt[2] + wgt[3] > 0.f)) continue;                 // an unskinned primitive of the mesh keeps its bind pose
                    const double p[3] = {out[3 * (size_t)v], out[3 * (size_t)v + 1], out[3 * (size_t)v + 2]}; double q[3] = {0, 0, 0};      // (the morphed position)
                    for (int k = 0; k < 4; k++) { if (wgt[k] == 0.f || jnt[k] >= jm.size()) continue; const double* m = jm[jnt[k]].m;
                        for (int rr = 0; rr < 3; rr++) q[rr] += (double)wgt[k] * (m[rr] * p[0] + m[4 + rr] * p[1] + m[8 + rr] * p[2] + m[12 + rr]); }
                    for (int rr = 0; rr < 3; rr++) out[3 * (size_t)v + rr] = (float)q[rr];
                    if ((outN && (gd.flags & PT_GEOM_HAS_NORMAL)) || (outT && (gd.flags & PT_GEOM_HAS_TANGENT))) {
                        double nrm[3] = {0, 0, 0}, tan[3] = {0, 0, 0}; const float* n0 = &nF[3 * (size_t)v]; const float* t0 = &tF[4 * (size_t)v];      // (morphed first)
                        for (int k = 0; k < 4; k++) { if (wgt[k] == 0.f || jnt[k] >= jm.size()) continue; const double* m = jm[jnt[k]].m;      // column major: m[4 c + r]
                            // cofactors of the upper 3 x 3 = det * inverse-transpose
                            const double c00 = m[5] * m[10] - m[9] * m[6], c01 = m[9] * m[2] - m[1] * m[10], c02 = m[1] * m[6] - m[5] * m[2];      // (rows of the cofactor matrix, indexed [row][col] of M)
                            const double det = m[0] * c00 + m[4] * c01 + m[8] * c02;
                            if (det != 0.0) {
                                const double C[3][3] = {{c00, c01, c02},
                                                        {m[8] * m[6] - m[4] * m[10], m[0] * m[10] - m[8] * m[2], m[4] * m[2] - m[0] * m[6]},
                                                        {m[4] * m[9] - m[8] * m[5], m[8] * m[1] - m[0] * m[9], m[0] * m[5] - m[4] * m[1]}};      // C[r][c]: cofactor of M(r, c), M(r, c) = m[4 c + r]; inverse-transpose = C / det
                                for (int rr = 0; rr < 3; rr++) nrm[rr] += (double)wgt[k] * (C[rr][0] * n0[0] + C[rr][1] * n0[1] + C[rr][2] * n0[2]) / det;
                            }
                            for (int rr = 0; rr < 3; rr++) tan[rr] += (double)wgt[k] * (m[rr] * t0[0] + m[4 + rr] * t0[1] + m[8 + rr] * t0[2]);
                        }
                        if (outN && (gd.flags & PT_GEOM_HAS_NORMAL)) { const double l = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
                            if (l > 0) { float nn[3] = {(float)(nrm[0] / l), (float)(nrm[1] / l), (float)(nrm[2] / l)}; outN[v] = pack_snorm8(nn, 3); } }
                        if (outT && (gd.flags & PT_GEOM_HAS_TANGENT)) { const double l = sqrt(tan[0] * tan[0] + tan[1] * tan[1] + tan[2] * tan[2]);
                            if (l > 0) { float tt[4] = {(float)(tan[0] / l), (float)(tan[1] / l), (float)(tan[2] / l), t0[3]}; outT[v] = pack_snorm8(tt, 4); } }
                    }
                }
            }
        }
        return (int32_t)nv;
    } catch (...) { a->L.recordWorlds = false; a->L.nodeOverride = nullptr; return -PT_ERROR_IO; }
}
extern "C" int32_t pt_gltf_animation_positions(pt_gltf_animation* a, uint32_t animation, float t, float* out, uint32_t capacityVertices) {
    if (capacityVertices && !out) return -PT_ERROR_INVALID_ARGUMENT;
    return gltf_animation_pose(a, animation, t, out, nullptr, nullptr, capacityVertices);
}
extern "C" int32_t pt_gltf_animation_normals(pt_gltf_animation* a, uint32_t animation, float t, uint32_t* normals, uint32_t* tangents, uint32_t capacityVertices) {
    if (capacityVertices && !normals && !tangents) return -PT_ERROR_INVALID_ARGUMENT;
    return gltf_animation_pose(a, animation, t, nullptr, normals, tangents, capacityVertices);
}
extern "C" int32_t pt_gltf_animation_instances(pt_gltf_animation* a, uint32_t animation, float t, PtInstanceDesc* out, uint32_t capacity) {
    if (!a || (capacity && !out)) return -PT_ERROR_INVALID_ARGUMENT;
    try {
        const JValue* nodes = a->L.root.get("nodes"); const size_t nn = nodes ? nodes->size() : 0;
        std::vector<Loader::NodeTRS> ov(nn); for (auto& o : ov) { o.has[0] = o.has[1] = o.has[2] = false; }
        if (animation < a->anims.size()) for (auto& c : a->anims[animation].channels) {
            if ((size_t)c.node >= nn || c.path > 2) continue;
            Loader::NodeTRS& o = ov[(size_t)c.node]; double v[4] = {0, 0, 0, 1};
            sample_channel(a->anims[animation].samplers[(size_t)c.sampler], c.path, (double)t, v);
            o.has[c.path] = true; if (c.path == 0) memcpy(o.t, v, 24); else if (c.path == 1) memcpy(o.q, v, 32); else memcpy(o.s, v, 24);
        }
        Loader& L = a->L; L.instances.clear(); L.instanceWorld.clear(); L.instancePath.clear(); L.nodeOverride = &ov;
        int sceneIdx = L.root.intOr("scene", 0); const JValue* scenes = L.root.get("scenes");
        if (scenes && (size_t)sceneIdx < scenes->size()) { if (const JValue* ns = scenes->arr[sceneIdx].get("nodes")) for (auto& n : ns->arr) L.visit((int)n.num, m4_identity(), 0); }
        else for (int r : gltf_parentless_nodes(nodes)) L.visit(r, m4_identity(), 0);
        L.nodeOverride = nullptr;
        const size_t n = L.instances.size() < capacity ? L.instances.size() : capacity;
        if (n) memcpy(out, L.instances.data(), n * sizeof(PtInstanceDesc));
        return (int32_t)L.instances.size();
    } catch (...) { a->L.nodeOverride = nullptr; return -PT_ERROR_IO; }
}

// ================================================================ RTXPT `.scene.json` asset folders
// ExtendedScene leaves (Rtxpt/SampleCommon/ExtendedScene.cpp:104-143, 313-375), what Sample::SceneLoaded does with them (Rtxpt/Sample.cpp:457-479,
// 520-640) and MaterialsBaker::Load's override lookup (Rtxpt/Materials/MaterialsBaker.cpp:707-748, 857-864). The graph format and the Donut light /
// camera keys are restated from Donut's published Scene.cpp / SceneGraph.cpp (not vendored in the reference tree); Sample::SaveCurrentCamera
// (Rtxpt/Sample.cpp:925-962) shows the same keys from the reference's side.
struct pt_scene_import {
    std::vector<uint32_t> indices; std::vector<float> positions, uvs; std::vector<uint32_t> normals, tangents;
    std::vector<PtGeometryDesc> geoms; std::vector<PtMeshDesc> meshes; std::vector<PtInstanceDesc> instances; std::vector<PTMaterialData> materials;
    std::vector<std::vector<uint8_t>> texPixels; std::vector<PtTextureDesc> texDescs;
    std::vector<PolymorphicLightInfo> lights; std::vector<PolymorphicLightInfoEx> lightsEx; std::vector<PtSceneCameraDesc> cameras;
    std::vector<PtEnvDirectionalLight> directionalLights;
    PtSceneJsonInfo info;
};

namespace {
bool jvec(const JValue* v, double* out, int n) {               // Donut's `>>` for vectors: an array of n numbers, or one number broadcast
    if (!v) return false;
    if (v->type == JValue::Num) { for (int i = 0; i < n; i++) out[i] = v->num; return true; }
    if (v->type != JValue::Arr || (int)v->arr.size() != n) return false;
    for (int i = 0; i < n; i++) if (v->arr[i].type != JValue::Num) return false;
    for (int i = 0; i < n; i++) out[i] = v->arr[i].num;
    return true;
}
std::string file_stem(const std::string& path) {
    size_t slash = path.find_last_of('/'); std::string f = slash == std::string::npos ? path : path.substr(slash + 1);
    size_t dot = f.find_last_of('.'); return dot == std::string::npos ? f : f.substr(0, dot);
}
struct ModelSlot { bool loaded = false; int32_t status = PT_OK; uint32_t firstMesh = 0; std::vector<int> meshRemap; std::vector<uint32_t> instMesh; std::vector<M4> instWorld; std::vector<std::string> instPath;
                   std::vector<Loader::PunctualRaw> punctual; std::vector<Loader::CameraRaw> cameras; };      // punctual: the model file's KHR_lights_punctual lights (in its own space)

struct SceneReader {
    pt_scene_import& S; std::string sceneDir, mediaDir, sceneStem; std::vector<std::string> modelPaths; std::vector<ModelSlot> slots; int32_t err = PT_OK;
    explicit SceneReader(pt_scene_import& s) : S(s) {}

    // PTMaterial::Read's loadTexture (MaterialsBaker.cpp:166-193): a `.dds` next to a `.png` wins (the reference's compression script puts BC7 files there); the sRGB flag
    // comes from the material document, not from the file
    uint32_t add_png_texture(const std::string& file, bool srgb) {
        std::vector<uint8_t> bytes; uint32_t w, h; std::vector<uint8_t> rgba;
        std::string ext = file.size() >= 4 ? file.substr(file.size() - 4) : std::string(); for (char& ch : ext) ch = (char)tolower((unsigned char)ch);
        std::string dds = ext == ".dds" ? file : std::string();
        if (ext == ".png") { std::string cand = file.substr(0, file.size() - 4) + ".dds"; if (FILE* f = fopen(cand.c_str(), "rb")) { fclose(f); dds = cand; } }
        if (!dds.empty()) {
            uint32_t fmt = 0; void* px = nullptr;
            if (pt_image_read_dds(dds.c_str(), &w, &h, &fmt, &px) != PT_OK) return 0xFFFFFFFFu;
            if (fmt == PT_TEX_RGBA32F) { pt_image_free((float*)px); return 0xFFFFFFFFu; }              // (float textures are environment sources, not material inputs)
            { PxGuard guard{px}; rgba.assign((const uint8_t*)px, (const uint8_t*)px + (size_t)w * h * 4u); }
        }
        else {
            if (!read_file(file, bytes)) return 0xFFFFFFFFu;
            if (bytes.size() > 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) { void* px = nullptr; if (pt_image_read_jpeg(bytes.data(), bytes.size(), &w, &h, &px) != PT_OK) return 0xFFFFFFFFu;
                                                                              { PxGuard guard{px}; rgba.assign((const uint8_t*)px, (const uint8_t*)px + (size_t)w * h * 4u); } }
            else if (!decode_png(bytes, w, h, rgba)) return 0xFFFFFFFFu;
        }
        uint32_t index = (uint32_t)S.texDescs.size(); S.texPixels.push_back(std::move(rgba));
        PtTextureDesc d; d.width = w; d.height = h; d.format = srgb ? PT_TEX_RGBA8_SRGB : PT_TEX_RGBA8_UNORM; d.pixels = nullptr; S.texDescs.push_back(d);
        return pack_texture_word(index, w, h);
    }
    // MaterialsBaker::Load: the four candidates in order; returns false when there is no document for this material
    bool material_override(const std::string& modelName, const std::string& name, PTMaterialData& out, PtMaterialJsonInfo& mi) {
        if (name.empty()) return false;
        const std::string base = mediaDir + "Materials/", ext = ".material.json";
        const std::string cand[4] = {base + sceneStem + "/" + modelName + "." + name + ext, base + sceneStem + "/" + name + ext, base + modelName + "." + name + ext, base + name + ext};
        for (int c = 0; c < 4; c++) {
            std::vector<uint8_t> text;
            if (!read_file(cand[c], text)) continue;
            text.push_back(0);
            PTMaterialData probe;
            if (pt_material_from_json((const char*)text.data(), nullptr, &probe, &mi) != PT_OK) continue;        // LoadJsonFromFile failed: next candidate
            uint32_t words[5];
            for (int t = 0; t < 5; t++) {
                words[t] = 0xFFFFFFFFu;
                if (!mi.texturePath[t][0]) continue;
                words[t] = add_png_texture(mediaDir + mi.texturePath[t], mi.textureSRGB[t] != 0);
                if (words[t] == 0xFFFFFFFFu) S.info.texturesNotLoaded++;
            }
            if (pt_material_from_json((const char*)text.data(), words, &out, &mi) != PT_OK) return false;
            return true;
        }
        return false;
    }
    // load models[k] once: append its textures, materials (with overrides), geometry and meshes; remember its instances for every node that uses it
    ModelSlot& model(size_t k) {
        ModelSlot& slot = slots[k];
        if (slot.loaded) return slot;
        slot.loaded = true;
        Loader L; L.ctx = nullptr;
        std::string path = sceneDir + modelPaths[k];
        slot.status = load_gltf_file(path.c_str(), L);
        if (slot.status != PT_OK) return slot;
        const uint32_t texOffset = (uint32_t)S.texDescs.size(), matOffset = (uint32_t)S.materials.size();
        for (size_t i = 0; i < L.texDescs.size(); i++) { S.texDescs.push_back(L.texDescs[i]); S.texPixels.push_back(std::move(L.texPixels[i])); }
        const JValue* mats = L.root.get("materials"); const std::string modelName = file_stem(modelPaths[k]);
        std::vector<int> geomFlagOverride(L.materials.size(), -1); std::vector<bool> skip(L.materials.size(), false);
        for (size_t i = 0; i < L.materials.size(); i++) {
            PTMaterialData m = L.materials[i];
            uint32_t* words[6] = {&m.BaseOrDiffuseTextureIndex, &m.MetalRoughOrSpecularTextureIndex, &m.EmissiveTextureIndex, &m.NormalTextureIndex, &m.OcclusionTextureIndex, &m.TransmissionTextureIndex};
            for (auto w : words) if (*w != 0xFFFFFFFFu) *w = (*w & 0xFFFF0000u) | ((*w & 0xFFFFu) + texOffset);
            std::string name = (mats && i < mats->size()) ? mats->arr[i].strOr("name", "") : std::string();
            PTMaterialData ov; PtMaterialJsonInfo mi;
            if (material_override(modelName, name, ov, mi)) {
                m = ov; S.info.materialOverrides++;
                geomFlagOverride[i] = (mi.enableAlphaTesting ? PT_GEOMF_ALPHA_TESTED : 0) | (mi.excludeFromNEE ? PT_GEOMF_EXCLUDE_FROM_NEE : 0);
                skip[i] = mi.skipRender != 0;
            }
            S.materials.push_back(m);
        }
        const uint32_t indexOffset = (uint32_t)S.indices.size(), vertexOffset = (uint32_t)(S.positions.size() / 3);
        S.indices.insert(S.indices.end(), L.indices.begin(), L.indices.end());
        S.positions.insert(S.positions.end(), L.positions.begin(), L.positions.end()); S.uvs.insert(S.uvs.end(), L.uvs.begin(), L.uvs.end());
        S.normals.insert(S.normals.end(), L.normals.begin(), L.normals.end()); S.tangents.insert(S.tangents.end(), L.tangents.begin(), L.tangents.end());
        slot.firstMesh = (uint32_t)S.meshes.size(); slot.meshRemap.assign(L.meshes.size(), -1);
        for (size_t mi = 0; mi < L.meshes.size(); mi++) {
            PtMeshDesc md; md.firstGeometry = (uint32_t)S.geoms.size(); md.numGeometries = 0;
            for (uint32_t g = 0; g < L.meshes[mi].numGeometries; g++) {
                PtGeometryDesc gd = L.geoms[L.meshes[mi].firstGeometry + g];
                if (skip[gd.materialIndex]) { S.info.skippedGeometries++; continue; }                    // SkipRender: the reference gives these a NaN transform (AccelerationStructureUtil.h:62-72)
                if (geomFlagOverride[gd.materialIndex] >= 0) gd.geomFlags = (uint32_t)geomFlagOverride[gd.materialIndex];
                gd.indexOffset += indexOffset; gd.vertexOffset += vertexOffset; gd.materialIndex += matOffset;
                S.geoms.push_back(gd); md.numGeometries++;
            }
            if (md.numGeometries) { slot.meshRemap[mi] = (int)S.meshes.size(); S.meshes.push_back(md); }
        }
        for (size_t i = 0; i < L.instances.size(); i++) {
            int mesh = slot.meshRemap[L.instances[i].meshIndex];
            if (mesh < 0) continue;
            slot.instMesh.push_back((uint32_t)mesh); slot.instWorld.push_back(L.instanceWorld[i]); slot.instPath.push_back(i < L.instancePath.size() ? L.instancePath[i] : std::string());
        }
        slot.punctual = L.punctual; S.info.lightsDropped += L.punctualDropped; slot.cameras = L.cameras;
        S.info.numModels++;
        return slot;
    }
    // Scene-graph paths (Donut SceneGraph::FindNode, un-vendored: restated from its published source, UNPINNED): node names joined by '/', a leading '/' = from the root.
    // A model hangs below its graph node as a root node named after the model file, the glTF scene's nodes below that.
    std::vector<std::string> instancePaths;                        // parallel to S.instances
    struct PendingProxy { uint32_t light; std::string path; }; std::vector<PendingProxy> pendingProxies;
    static std::string file_name_of(const std::string& p) { size_t k = p.find_last_of("/\\"); return k == std::string::npos ? p : p.substr(k + 1); }
    void instantiate(size_t k, const M4& world, const std::string& nodePath = std::string()) {
        if (k >= modelPaths.size()) { err = PT_ERROR_IO; return; }
        ModelSlot& slot = model(k);
        if (slot.status != PT_OK) { err = slot.status; return; }
        for (size_t i = 0; i < slot.instMesh.size(); i++) {
            instancePaths.push_back(nodePath + "/" + file_name_of(modelPaths[k]) + (i < slot.instPath.size() && !slot.instPath[i].empty() ? "/" + slot.instPath[i] : std::string()));
            M4 w = m4_mul(world, slot.instWorld[i]);
            PtInstanceDesc inst; memset(&inst, 0, sizeof(inst)); inst.meshIndex = slot.instMesh[i];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) inst.transform[r * 4 + c] = (float)w.m[c * 4 + r];
            S.instances.push_back(inst);
        }
        // the model's KHR_lights_punctual lights hang below the model node like its meshes: LightsBaker's list / the environment baker's list grow in scene-graph order
        std::vector<PtAnalyticLightDesc> analytic; std::vector<PtEnvDirectionalLight> directional;
        for (const Loader::PunctualRaw& l : slot.punctual) Loader::emit_punctual(l, world, analytic, directional);
        for (const PtAnalyticLightDesc& d : analytic) { PolymorphicLightInfo b; PolymorphicLightInfoEx e; int32_t r = pt_convert_light(&d, &b, &e); if (r != PT_OK) { err = r; return; } S.lights.push_back(b); S.lightsEx.push_back(e); }
        for (const PtEnvDirectionalLight& d : directional) { S.directionalLights.push_back(d); S.info.directionalLights++; }
        for (const Loader::CameraRaw& c : slot.cameras) S.cameras.push_back(Loader::emit_camera(c, world));       // the model's own cameras, in scene-graph order with the graph's
    }
    // ExtendedScene::ProcessNodesRecursive (ExtendedScene.cpp:246-263) + LightsBaker::Update (LightsBaker.cpp:718-753): the mesh instance a point / spot light names in
    // "proxyMeshNodes" stands in for that light (PtInstanceDesc.analyticProxyLight). Only a path that ends at a node holding a mesh instance links, as there.
    void resolve_proxies() {
        for (const PendingProxy& pp : pendingProxies) {
            std::string want = pp.path; while (!want.empty() && want[0] == '/') want.erase(0, 1);
            std::vector<std::string> parts; { size_t a = 0; while (a <= want.size()) { size_t b = want.find('/', a); if (b == std::string::npos) b = want.size(); std::string c = want.substr(a, b - a); if (c == "..") { if (!parts.empty()) parts.pop_back(); } else if (!c.empty() && c != ".") parts.push_back(c); a = b + 1; } }
            std::string norm; for (auto& c : parts) norm += "/" + c;
            for (size_t i = 0; i < instancePaths.size() && i < S.instances.size(); i++) if (instancePaths[i] == norm) { S.instances[i].analyticProxyLight = pp.light + 1u; S.info.lightProxiesResolved++; }
        }
    }
    void leaf(const JValue& n, const std::string& type, const M4& world) {
        PtSceneJsonInfo& I = S.info;
        const double zx = world.m[8], zy = world.m[9], zz = world.m[10];                    // image of the local Z axis
        if (type == "EnvironmentLight") {                                                    // ExtendedScene.cpp:104-110; the first one wins (FindEnvironmentLight :295-305)
            if (I.hasEnvironment) return;
            I.hasEnvironment = 1; I.envRadianceScale[0] = I.envRadianceScale[1] = I.envRadianceScale[2] = 1.f; I.envTextureIndex = -1; I.envRotation = 0.f;
            double v[3]; if (jvec(n.get("radianceScale"), v, 3)) for (int i = 0; i < 3; i++) I.envRadianceScale[i] = (float)v[i];
            const JValue* j = n.get("textureIndex"); if (j && j->type == JValue::Num) I.envTextureIndex = (int32_t)j->num;
            j = n.get("rotation"); if (j && j->type == JValue::Num) I.envRotation = (float)j->num;
            std::string p = n.strOr("path", ""); strncpy(I.envPath, p.c_str(), sizeof(I.envPath) - 1);
        } else if (type == "PointLight" || type == "SpotLight") {
            PtAnalyticLightDesc d; memset(&d, 0, sizeof(d));
            d.type = type == "SpotLight" ? 1u : 0u; d.color[0] = d.color[1] = d.color[2] = 1.f; d.intensity = 1.f; d.radius = 0.f; d.innerAngle = 180.f; d.outerAngle = 180.f;   // Donut defaults
            double v[3]; if (jvec(n.get("color"), v, 3)) for (int i = 0; i < 3; i++) d.color[i] = (float)v[i];
            jload(n, "intensity", d.intensity); jload(n, "radius", d.radius);
            if (d.type == 1u) { jload(n, "innerAngle", d.innerAngle); jload(n, "outerAngle", d.outerAngle); }
            const JValue* px = n.get("proxyMeshNodes");
            if (px) I.lightProxies += (uint32_t)px->size();
            float cx = d.color[0] * d.intensity, cy = d.color[1] * d.intensity, cz = d.color[2] * d.intensity;
            if (sqrtf(cx * cx + cy * cy + cz * cz) <= 1e-7f) { I.lightsDropped++; return; }     // Sample.cpp:567-573
            double len = sqrt(zx * zx + zy * zy + zz * zz); if (!(len > 0)) len = 1;
            for (int i = 0; i < 3; i++) d.position[i] = (float)world.m[12 + i];
            d.direction[0] = (float)(-zx / len); d.direction[1] = (float)(-zy / len); d.direction[2] = (float)(-zz / len);      // Light::GetDirection: -Z of the node
            PolymorphicLightInfo b; PolymorphicLightInfoEx e;
            int32_t r = pt_convert_light(&d, &b, &e);
            if (r != PT_OK) { err = r; return; }
            if (px) for (auto& q : px->arr) if (q.type == JValue::Str) pendingProxies.push_back({(uint32_t)S.lights.size(), q.str});      // JsonLoadStringVector: the strings of the array
            S.lights.push_back(b); S.lightsEx.push_back(e);
        } else if (type == "DirectionalLight") {                                             // not part of LightsBaker's light set (LightsBaker.cpp:600): baked into the environment cube
            I.directionalLights++;                                                           // (Sample::UpdateLighting, Sample.cpp:1361-1388). Donut keys: color, irradiance, angularSize [deg]
            PtEnvDirectionalLight d; memset(&d, 0, sizeof(d));
            float color[3] = {1.f, 1.f, 1.f}, irradiance = 1.f, angularSize = 0.f;
            double v[3]; if (jvec(n.get("color"), v, 3)) for (int i = 0; i < 3; i++) color[i] = (float)v[i];
            jload(n, "irradiance", irradiance); jload(n, "angularSize", angularSize);
            float cx = color[0] * irradiance, cy = color[1] * irradiance, cz = color[2] * irradiance;
            if (sqrtf(cx * cx + cy * cy + cz * cz) <= 1e-7f) { I.lightsDropped++; return; }     // Sample.cpp:567-573
            double len = sqrt(zx * zx + zy * zy + zz * zz); if (!(len > 0)) len = 1;
            d.ColorIntensity[0] = color[0]; d.ColorIntensity[1] = color[1]; d.ColorIntensity[2] = color[2]; d.ColorIntensity[3] = irradiance;      // DirectionalLight::FillLightConstants
            d.Direction[0] = (float)(-zx / len); d.Direction[1] = (float)(-zy / len); d.Direction[2] = (float)(-zz / len);
            d.AngularSize = std::min(std::max(angularSize, 0.f), 90.f) * (3.141592654f / 180.f);
            S.directionalLights.push_back(d);
        }
        else if (type == "PerspectiveCamera" || type == "PerspectiveCameraEx") {           // ExtendedScene.cpp:329-338 + Sample::UpdateCameraFromScene
            PtSceneCameraDesc c; memset(&c, 0, sizeof(c)); c.verticalFov = 1.f; c.zNear = 1.f;
            jload(n, "verticalFov", c.verticalFov); jload(n, "zNear", c.zNear);
            for (int i = 0; i < 3; i++) { c.position[i] = (float)world.m[12 + i]; c.up[i] = (float)world.m[4 + i]; }
            c.direction[0] = (float)-zx; c.direction[1] = (float)-zy; c.direction[2] = (float)-zz;       // row 2 of scaling(1,1,-1) * localToWorld
            bool b = false; if (const JValue* j = n.get("enableAutoExposure")) if (j->type == JValue::Bool || j->type == JValue::Num) { jload(n, "enableAutoExposure", b); c.enableAutoExposure = b; c.exposureMask |= 1u; }
            static const char* keys[4] = {"exposureCompensation", "exposureValue", "exposureValueMin", "exposureValueMax"}; float* dst[4] = {&c.exposureCompensation, &c.exposureValue, &c.exposureValueMin, &c.exposureValueMax};
            for (int k = 0; k < 4; k++) if (const JValue* j = n.get(keys[k])) if (j->type == JValue::Num) { *dst[k] = (float)j->num; c.exposureMask |= 2u << k; }
            std::string name = n.strOr("name", ""); strncpy(c.name, name.c_str(), sizeof(c.name) - 1);
            S.cameras.push_back(c);
        } else if (type == "SampleSettings") {                                               // ExtendedScene.cpp:347-356 (the last node wins, :243-247)
            I.settingsMask = 0;
            bool b;
            if (const JValue* j = n.get("realtimeMode")) if (j->type == JValue::Bool || j->type == JValue::Num) { b = false; jload(n, "realtimeMode", b); I.realtimeMode = b; I.settingsMask |= 1u; }
            if (const JValue* j = n.get("enableAnimations")) if (j->type == JValue::Bool || j->type == JValue::Num) { b = false; jload(n, "enableAnimations", b); I.enableAnimations = b; I.settingsMask |= 2u; }
            if (const JValue* j = n.get("startingCamera")) if (j->type == JValue::Num) { I.startingCamera = (int32_t)j->num; I.settingsMask |= 4u; }
            if (const JValue* j = n.get("realtimeFireflyFilter")) if (j->type == JValue::Num) { I.realtimeFireflyFilter = (float)j->num; I.settingsMask |= 8u; }
            if (const JValue* j = n.get("maxBounces")) if (j->type == JValue::Num) { I.maxBounces = (int32_t)j->num; I.settingsMask |= 16u; }
            if (const JValue* j = n.get("maxDiffuseBounces")) if (j->type == JValue::Num) { I.maxDiffuseBounces = (int32_t)j->num; I.settingsMask |= 32u; }
            if (const JValue* j = n.get("textureMIPBias")) if (j->type == JValue::Num) { I.textureMIPBias = (float)j->num; I.settingsMask |= 64u; }
        }
        // GameSettings, unknown types: nothing the path tracer consumes
    }
    void node(const JValue& n, const M4& parent, int depth, const std::string& parentPath = std::string()) {
        if (err != PT_OK || depth > 256) return;
        if (n.type == JValue::Str) {                                                         // a bare string names a model
            for (size_t k = 0; k < modelPaths.size(); k++) if (modelPaths[k] == n.str) { instantiate(k, parent, parentPath); return; }
            err = PT_ERROR_IO; return;
        }
        if (n.type != JValue::Obj) return;
        double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
        jvec(n.get("translation"), t, 3); jvec(n.get("scaling"), s, 3);
        if (const JValue* r = n.get("rotation")) { if (r->type == JValue::Arr) jvec(r, q, 4); }     // (EnvironmentLight's scalar "rotation" is a leaf property, not a node rotation)
        else if (const JValue* e = n.get("euler")) {
            // Donut's dm::rotationQuat(euler) (core/math/quat.h; un-vendored — GameMisc.cpp:53-61 shows the reference's own use of it): half-angle quaternions about x, y and z
            // multiplied qZ * qY * qX, i.e. the rotation about the fixed x axis is applied first, then y, then z. UNPINNED (restated from Donut's published header).
            double a[3] = {0, 0, 0}; jvec(e, a, 3);
            const double cx = cos(0.5 * a[0]), sx = sin(0.5 * a[0]), cy = cos(0.5 * a[1]), sy = sin(0.5 * a[1]), cz = cos(0.5 * a[2]), sz = sin(0.5 * a[2]);
            auto mul = [](const double* p, const double* r, double* o) {      // Hamilton product, (w, x, y, z)
                o[0] = p[0] * r[0] - p[1] * r[1] - p[2] * r[2] - p[3] * r[3]; o[1] = p[0] * r[1] + p[1] * r[0] + p[2] * r[3] - p[3] * r[2];
                o[2] = p[0] * r[2] - p[1] * r[3] + p[2] * r[0] + p[3] * r[1]; o[3] = p[0] * r[3] + p[1] * r[2] - p[2] * r[1] + p[3] * r[0]; };
            const double qx[4] = {cx, sx, 0, 0}, qy[4] = {cy, 0, sy, 0}, qz[4] = {cz, 0, 0, sz}; double zy[4], w[4];
            mul(qz, qy, zy); mul(zy, qx, w);
            q[0] = w[1]; q[1] = w[2]; q[2] = w[3]; q[3] = w[0];
        }
        M4 world = m4_mul(parent, m4_trs(t, q, s));
        const std::string path = parentPath + "/" + n.strOr("name", "");
        if (const JValue* m = n.get("model")) {
            if (m->type == JValue::Num) instantiate((size_t)m->num, world, path);
            else if (m->type == JValue::Str) { bool found = false; for (size_t k = 0; k < modelPaths.size(); k++) if (modelPaths[k] == m->str) { instantiate(k, world, path); found = true; break; } if (!found) err = PT_ERROR_IO; }
        }
        if (const JValue* ty = n.get("type")) if (ty->type == JValue::Str) leaf(n, ty->str, world);
        if (const JValue* ch = n.get("children")) for (auto& c : ch->arr) node(c, world, depth + 1, path);
    }
};
} // namespace

static int32_t scene_json_import_impl(const char* scenePath, const char* mediaPath, pt_scene_import** out, PtSceneJsonInfo* info);
extern "C" int32_t pt_scene_json_import(const char* scenePath, const char* mediaPath, pt_scene_import** out, PtSceneJsonInfo* info) {
    if (!scenePath || !out) return PT_ERROR_INVALID_ARGUMENT;
    *out = nullptr;
    try { return scene_json_import_impl(scenePath, mediaPath, out, info); } catch (...) { *out = nullptr; return PT_ERROR_IO; }
}
static int32_t scene_json_import_impl(const char* scenePath, const char* mediaPath, pt_scene_import** out, PtSceneJsonInfo* info) {
    std::vector<uint8_t> file;
    if (!read_file(scenePath, file)) return PT_ERROR_IO;
    JParser jp; jp.p = (const char*)file.data(); jp.e = jp.p + file.size(); JValue root = jp.parse();
    if (!jp.ok || root.type != JValue::Obj) return PT_ERROR_IO;
    std::unique_ptr<pt_scene_import> S(new pt_scene_import()); memset(&S->info, 0, sizeof(S->info)); S->info.selectedCamera = -1; S->info.envTextureIndex = -1;
    SceneReader R(*S);
    std::string sp(scenePath); size_t slash = sp.find_last_of('/'); R.sceneDir = slash == std::string::npos ? "" : sp.substr(0, slash + 1);
    R.mediaDir = mediaPath ? std::string(mediaPath) : R.sceneDir; if (!R.mediaDir.empty() && R.mediaDir.back() != '/') R.mediaDir += '/';
    R.sceneStem = file_stem(sp);                                                             // "bistro.scene.json" -> "bistro.scene": filename().stem() strips one extension (MaterialsBaker.cpp:862)
    if (const JValue* models = root.get("models")) for (auto& m : models->arr) { if (m.type != JValue::Str) return PT_ERROR_IO; R.modelPaths.push_back(m.str); }
    R.slots.resize(R.modelPaths.size());
    if (const JValue* graph = root.get("graph")) for (auto& n : graph->arr) R.node(n, m4_identity(), 0);
    R.resolve_proxies();
    if (R.err != PT_OK) return R.err;
    PtSceneJsonInfo& I = S->info;
    I.numGeometries = (uint32_t)S->geoms.size(); I.numMeshes = (uint32_t)S->meshes.size(); I.numInstances = (uint32_t)S->instances.size(); I.numMaterials = (uint32_t)S->materials.size();
    I.numTextures = (uint32_t)S->texDescs.size(); I.numLights = (uint32_t)S->lights.size(); I.numCameras = (uint32_t)S->cameras.size();
    if (I.numCameras) I.selectedCamera = ((I.settingsMask & 4u) && I.startingCamera >= 0 && (uint32_t)I.startingCamera < I.numCameras) ? I.startingCamera : (int32_t)I.numCameras - 1;
    for (size_t i = 0; i < S->texDescs.size(); i++) S->texDescs[i].pixels = S->texPixels[i].data();
    if (info) *info = I;
    *out = S.release();
    return PT_OK;
}
extern "C" void pt_scene_import_free(pt_scene_import* scene) { delete scene; }
#define PT_IMPORT_COPY(FN, TYPE, VEC) \
    extern "C" int32_t FN(const pt_scene_import* scene, TYPE* out, uint32_t capacity) { \
        if (!scene || (capacity && !out)) return -PT_ERROR_INVALID_ARGUMENT; \
        size_t n = scene->VEC.size() < capacity ? scene->VEC.size() : capacity; \
        if (n) memcpy(out, scene->VEC.data(), n * sizeof(TYPE)); \
        return (int32_t)scene->VEC.size(); }
PT_IMPORT_COPY(pt_scene_import_cameras, PtSceneCameraDesc, cameras)
PT_IMPORT_COPY(pt_scene_import_instances, PtInstanceDesc, instances)
PT_IMPORT_COPY(pt_scene_import_geometries, PtGeometryDesc, geoms)
PT_IMPORT_COPY(pt_scene_import_materials, PTMaterialData, materials)
#undef PT_IMPORT_COPY
extern "C" int32_t pt_scene_import_vertices(const pt_scene_import* scene, float* positions, uint32_t capacityVertices) {      // the imported POSITION stream (object space, xyz per vertex)
    if (!scene || (capacityVertices && !positions)) return -PT_ERROR_INVALID_ARGUMENT;
    const size_t nv = scene->positions.size() / 3, n = nv < capacityVertices ? nv : capacityVertices;
    if (n) memcpy(positions, scene->positions.data(), n * 12u);
    return (int32_t)nv;
}
extern "C" int32_t pt_scene_import_texture(const pt_scene_import* scene, uint32_t index, PtTextureDesc* out) {
    if (!scene || !out || index >= scene->texDescs.size()) return PT_ERROR_INVALID_ARGUMENT;
    *out = scene->texDescs[index]; out->pixels = scene->texPixels[index].data();
    return PT_OK;
}
extern "C" int32_t pt_scene_import_lights(const pt_scene_import* scene, PolymorphicLightInfo* base, PolymorphicLightInfoEx* ex, uint32_t capacity) {
    if (!scene || (capacity && (!base || !ex))) return -PT_ERROR_INVALID_ARGUMENT;
    size_t n = scene->lights.size() < capacity ? scene->lights.size() : capacity;
    if (n) { memcpy(base, scene->lights.data(), n * sizeof(PolymorphicLightInfo)); memcpy(ex, scene->lightsEx.data(), n * sizeof(PolymorphicLightInfoEx)); }
    return (int32_t)scene->lights.size();
}
extern "C" int32_t pt_scene_import_directional_lights(const pt_scene_import* scene, PtEnvDirectionalLight* out, uint32_t capacity) {
    if (!scene || (capacity && !out)) return -PT_ERROR_INVALID_ARGUMENT;
    size_t n = scene->directionalLights.size() < capacity ? scene->directionalLights.size() : capacity;
    if (n) memcpy(out, scene->directionalLights.data(), n * sizeof(PtEnvDirectionalLight));
    return (int32_t)scene->directionalLights.size();
}
// Sample::SceneLoaded, Rtxpt/Sample.cpp:613-629: m_ui.BounceCount / DiffuseBounceCount / TexLODBias = value_or(current). realtimeMode, enableAnimations and
// realtimeFireflyFilter steer the realtime path and the animation clock, which the reference-mode settings block does not hold.
extern "C" int32_t pt_scene_import_settings(const pt_scene_import* S, PtSettings* settings) {
    if (!S || !settings) return PT_ERROR_INVALID_ARGUMENT;
    const PtSceneJsonInfo& I = S->info;
    if ((I.settingsMask & 16u) && I.maxBounces >= 0) settings->bounceCount = (uint32_t)I.maxBounces;
    if ((I.settingsMask & 32u) && I.maxDiffuseBounces >= 0) settings->diffuseBounceCount = (uint32_t)I.maxDiffuseBounces;
    if (I.settingsMask & 64u) settings->texLODBias = I.textureMIPBias;
    return PT_OK;
}
extern "C" int32_t pt_scene_import_tone_mapping(const pt_scene_import* S, int32_t cameraIndex, PtToneMappingParameters* ui) {
    if (!S || !ui) return PT_ERROR_INVALID_ARGUMENT;
    ui->exposureCompensation = 2.0f; ui->exposureValue = 0.0f;                                  // Sample.cpp:547-549
    if (cameraIndex < 0) cameraIndex = S->info.selectedCamera;
    if (cameraIndex < 0) return PT_OK;                                                          // no camera in the scene
    if ((size_t)cameraIndex >= S->cameras.size()) return PT_ERROR_INVALID_ARGUMENT;
    const PtSceneCameraDesc& c = S->cameras[(size_t)cameraIndex];                               // a camera leaf of the graph is a PerspectiveCameraEx (ExtendedScene.cpp:126-129);
    if (c.exposureMask & 0x80000000u) return PT_OK;                                             // a glTF file's own camera is a plain PerspectiveCamera: the cast fails, nothing is overwritten (Sample.cpp:465-466)
    ui->autoExposure = (c.exposureMask & 1u) ? c.enableAutoExposure : 0u;
    ui->exposureCompensation = (c.exposureMask & 2u) ? c.exposureCompensation : 0.0f;
    ui->exposureValue = (c.exposureMask & 4u) ? c.exposureValue : 0.0f;
    ui->exposureValueMin = (c.exposureMask & 8u) ? c.exposureValueMin : -16.0f;
    ui->exposureValueMax = (c.exposureMask & 16u) ? c.exposureValueMax : 16.0f;
    return PT_OK;
}
// pt_load_scene_gltf on a `.scene.json`: what Sample::SceneLoaded leaves behind as far as this library holds it (Sample.cpp:520-640) — materials, geometry, instances, analytic
// lights (pt_scene_import_apply), the graph's directional lights for the environment bake (Sample::UpdateLighting's inputs), and the EnvironmentLight's image: the reference
// hands its `path` (relative to the media folder) to EnvMapBaker, which loads a lat-long image or a cube map (EnvMapBaker.cpp:392-415) with the UI block at identity
// (Sample.cpp:554, 1936-1948): tint 1, intensity 1, no rotation. A file that cannot be read leaves the scene without an image, as there. Camera and SampleSettings stay with
// the caller (pt_scene_json_import + pt_scene_import_cameras / _settings / _tone_mapping read them).
static int32_t load_scene_json_into(pt_context* ctx, const char* path) {
    pt_scene_import* S = nullptr; PtSceneJsonInfo info;
    int32_t r = pt_scene_json_import(path, nullptr, &S, &info);
    if (r != PT_OK) return r;
    std::unique_ptr<pt_scene_import> hold(S);
    r = pt_scene_import_apply(ctx, S); if (r != PT_OK) return r;
    r = pt_set_scene_directional_lights(ctx, S->directionalLights.empty() ? nullptr : S->directionalLights.data(), (uint32_t)std::min<size_t>(S->directionalLights.size(), 16)); if (r != PT_OK) return r;
    if (info.hasEnvironment && info.envPath[0] && strncmp(info.envPath, "==", 2) != 0) {      // ("==PROCEDURAL_SKY==" and its presets name the procedural sky: pt_set_procedural_sky, the caller's)
        std::string sp(path); const size_t slash = sp.find_last_of('/'); const std::string file = (slash == std::string::npos ? std::string() : sp.substr(0, slash + 1)) + info.envPath;
        uint32_t w = 0, h = 0, dim = 0; float* px = nullptr;
        if (pt_image_read_float(file.c_str(), &w, &h, &px) == PT_OK) { r = pt_set_environment(ctx, px, w, h, nullptr); pt_image_free(px); }
        else if (pt_image_read_dds_cube(file.c_str(), &dim, &px) == PT_OK) { r = pt_set_environment_cube(ctx, px, dim, nullptr); pt_image_free(px); }
        else r = pt_set_environment(ctx, nullptr, 0, 0, nullptr);
        if (r != PT_OK) return r;
    }
    return PT_OK;
}
extern "C" int32_t pt_scene_import_apply(pt_context* ctx, const pt_scene_import* S) {
    if (!ctx || !S) return PT_ERROR_INVALID_ARGUMENT;
    if (S->instances.empty() || S->geoms.empty()) return PT_ERROR_IO;
    int32_t r = pt_set_materials(ctx, S->materials.data(), (uint32_t)S->materials.size(), S->texDescs.data(), (uint32_t)S->texDescs.size());
    if (r != PT_OK) return r;
    PtGeometryBuffers gb; gb.indices = S->indices.data(); gb.numIndices = (uint32_t)S->indices.size(); gb.positions = S->positions.data(); gb.uvs = S->uvs.data();
    gb.normals = S->normals.data(); gb.tangents = S->tangents.data(); gb.numVertices = (uint32_t)(S->positions.size() / 3);
    r = pt_set_geometry(ctx, &gb, S->geoms.data(), (uint32_t)S->geoms.size(), S->meshes.data(), (uint32_t)S->meshes.size());
    if (r != PT_OK) return r;
    r = pt_set_instances(ctx, S->instances.data(), (uint32_t)S->instances.size());
    if (r != PT_OK) return r;
    return pt_set_lights(ctx, S->lights.data(), S->lightsEx.data(), (uint32_t)S->lights.size());
}
