// ImageIO.cpp — see ImageIO.h
#include "ImageIO.h"
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256]; static bool init = false;
    if (!init) { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; } init = true; }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }
void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4); out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0xffffffffu, &out[at], 4 + data.size()) ^ 0xffffffffu);
}
bool write_file(const std::string& path, const void* p, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(p, 1, n, f) == n;
    return fclose(f) == 0 && ok;
}
template <class T> void put_le(std::vector<uint8_t>& v, T x) { uint8_t b[sizeof(T)]; memcpy(b, &x, sizeof(T)); v.insert(v.end(), b, b + sizeof(T)); }   // host is little-endian (x86-64)
void put_str(std::vector<uint8_t>& v, const char* s) { v.insert(v.end(), s, s + strlen(s) + 1); }
}  // namespace

bool WritePNG(const std::string& path, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    if (!rgba8 || !w || !h) return false;
    std::vector<uint8_t> raw; raw.reserve((size_t)h * (1 + (size_t)w * 4));
    for (uint32_t y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgba8 + (size_t)y * w * 4, rgba8 + (size_t)(y + 1) * w * 4); }   // filter 0 per scanline
    std::vector<uint8_t> z; z.push_back(0x78); z.push_back(0x01);                       // zlib header, then stored deflate blocks
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < raw.size(); i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    for (size_t at = 0; at < raw.size();) {
        const size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n); at += n;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr; put_be32(ihdr, w); put_be32(ihdr, h); ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr); put_chunk(out, "IDAT", z); put_chunk(out, "IEND", {});
    return write_file(path, out.data(), out.size());
}

bool WritePPM(const std::string& path, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    if (!rgba8 || !w || !h) return false;
    char hdr[64]; const int n = snprintf(hdr, sizeof hdr, "P6\n%u %u\n255\n", w, h);
    std::vector<uint8_t> out(hdr, hdr + n); out.reserve(n + (size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) out.insert(out.end(), rgba8 + i * 4, rgba8 + i * 4 + 3);
    return write_file(path, out.data(), out.size());
}

bool WriteEXR(const std::string& path, const float* acc, uint32_t w, uint32_t h) {
    if (!acc || !w || !h) return false;
    std::vector<uint8_t> o;
    put_le<uint32_t>(o, 20000630u); put_le<uint32_t>(o, 2u);                            // magic, version 2, single-part scanline
    put_str(o, "channels"); put_str(o, "chlist"); put_le<uint32_t>(o, 3 * 18 + 1);
    for (const char* ch : {"B", "G", "R"}) { put_str(o, ch); put_le<uint32_t>(o, 2u /* FLOAT */); put_le<uint32_t>(o, 0u); put_le<uint32_t>(o, 1u); put_le<uint32_t>(o, 1u); }
    o.push_back(0);
    put_str(o, "compression"); put_str(o, "compression"); put_le<uint32_t>(o, 1u); o.push_back(0);
    const int32_t box[4] = {0, 0, (int32_t)w - 1, (int32_t)h - 1};
    for (const char* name : {"dataWindow", "displayWindow"}) { put_str(o, name); put_str(o, "box2i"); put_le<uint32_t>(o, 16u); for (int k = 0; k < 4; k++) put_le<int32_t>(o, box[k]); }
    put_str(o, "lineOrder"); put_str(o, "lineOrder"); put_le<uint32_t>(o, 1u); o.push_back(0);
    put_str(o, "pixelAspectRatio"); put_str(o, "float"); put_le<uint32_t>(o, 4u); put_le<float>(o, 1.0f);
    put_str(o, "screenWindowCenter"); put_str(o, "v2f"); put_le<uint32_t>(o, 8u); put_le<float>(o, 0.0f); put_le<float>(o, 0.0f);
    put_str(o, "screenWindowWidth"); put_str(o, "float"); put_le<uint32_t>(o, 4u); put_le<float>(o, 1.0f);
    o.push_back(0);                                                                     // end of header
    const size_t line_bytes = (size_t)w * 3 * 4, table_at = o.size();
    uint64_t off = table_at + (uint64_t)h * 8;
    for (uint32_t y = 0; y < h; y++) { put_le<uint64_t>(o, off); off += 8 + line_bytes; }
    for (uint32_t y = 0; y < h; y++) {
        put_le<int32_t>(o, (int32_t)y); put_le<uint32_t>(o, (uint32_t)line_bytes);
        for (int c = 2; c >= 0; c--)                                                    // channels in alphabetical order: B, G, R
            for (uint32_t x = 0; x < w; x++) { const float* p = acc + ((size_t)y * w + x) * 4; const float n = p[3] > 1.0f ? p[3] : 1.0f; put_le<float>(o, p[c] / n); }
    }
    return write_file(path, o.data(), o.size());
}
