// mi355pt — screenshot writers (host only): 8-bit RGBA PNG (zlib deflate, filter 0) and 32-bit BMP, the two lossless formats of the
// reference's SaveTextureToFile behind --captureSimple / --capturePath (Rtxpt/SampleCommon/CaptureScriptManager.cpp:29-60, Rtxpt/Sample.cpp:2295).
#include "../../include/mi355pt.h"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
void put_be32(std::vector<unsigned char>& v, uint32_t x) { v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x); }
void put_chunk(std::vector<unsigned char>& out, const char type[4], const unsigned char* data, size_t n) {
    put_be32(out, (uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4));
    put_be32(out, crc);
}
bool write_all(const char* path, const std::vector<unsigned char>& bytes) {
    FILE* f = fopen(path, "wb"); if (!f) return false;
    bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    ok = (fclose(f) == 0) && ok;
    return ok;
}
} // namespace

extern "C" int32_t pt_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!path || !rgba8 || !width || !height) return PT_ERROR_INVALID_ARGUMENT;
    std::vector<unsigned char> raw((size_t)height * ((size_t)width * 4 + 1));
    for (uint32_t y = 0; y < height; y++) {
        unsigned char* row = raw.data() + (size_t)y * ((size_t)width * 4 + 1);
        row[0] = 0;                                                       // filter type None
        memcpy(row + 1, rgba8 + (size_t)y * width * 4, (size_t)width * 4);
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) return PT_ERROR_IO;
    std::vector<unsigned char> out;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.insert(out.end(), sig, sig + 8);
    std::vector<unsigned char> ihdr; put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8 bit, RGBA, deflate, adaptive filtering, no interlace
    put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    static const unsigned char srgb[1] = {0};                            // sRGB chunk, perceptual intent: the bytes are sRGB-encoded (SRGBA8_UNORM target)
    put_chunk(out, "sRGB", srgb, 1);
    put_chunk(out, "IDAT", z.data(), (size_t)bound);
    put_chunk(out, "IEND", nullptr, 0);
    return write_all(path, out) ? PT_OK : PT_ERROR_IO;
}

extern "C" int32_t pt_write_bmp(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!path || !rgba8 || !width || !height) return PT_ERROR_INVALID_ARGUMENT;
    const uint32_t headerBytes = 14 + 40, imageBytes = width * height * 4;
    std::vector<unsigned char> out(headerBytes + (size_t)imageBytes, 0);
    auto le32 = [&](size_t at, uint32_t v) { out[at] = (unsigned char)v; out[at + 1] = (unsigned char)(v >> 8); out[at + 2] = (unsigned char)(v >> 16); out[at + 3] = (unsigned char)(v >> 24); };
    out[0] = 'B'; out[1] = 'M'; le32(2, headerBytes + imageBytes); le32(10, headerBytes);
    le32(14, 40); le32(18, width); le32(22, (uint32_t)(-(int32_t)height));      // negative height: rows top to bottom
    out[26] = 1; out[28] = 32; le32(34, imageBytes); le32(38, 2835); le32(42, 2835);
    for (size_t i = 0; i < (size_t)width * height; i++) {                        // BGRA
        out[headerBytes + 4 * i + 0] = rgba8[4 * i + 2]; out[headerBytes + 4 * i + 1] = rgba8[4 * i + 1];
        out[headerBytes + 4 * i + 2] = rgba8[4 * i + 0]; out[headerBytes + 4 * i + 3] = rgba8[4 * i + 3];
    }
    return write_all(path, out) ? PT_OK : PT_ERROR_IO;
}
