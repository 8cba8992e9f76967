// exr.h — minimal OpenEXR writer for Runtime::saveFramebuffer (src/runtime/Runtime.cpp:794-876, Image::save):
// single-part scanline file, 32-bit float channels, channels "B", "G", "R" like the reference's swizzle. Compressed like the
// reference's files for small films (Image.cpp:940-942: ZIP; it takes PIZ for larger ones -- both lossless for float, so the
// choice changes bytes on disk, not pixel values): ZIP_COMPRESSION, blocks of 16 scanlines, bytes split into even / odd halves,
// delta-predicted, deflated; a block that does not shrink is stored as is (the OpenEXR rule readers rely on).
// IGH_EXR_COMPRESSION=none writes the uncompressed form.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <zlib.h>

namespace igh {

inline void writeExr(const std::string& path, const float* rgb, int width, int height, float scale,
                     const std::vector<std::pair<std::string, std::string>>& string_attributes)
{
    if (width <= 0 || height <= 0 || !rgb)
        throw std::runtime_error("EXR '" + path + "': empty image");
    std::vector<uint8_t> out;
    auto put     = [&](const void* p, size_t n) { out.insert(out.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
    auto put_i32 = [&](int32_t v) { put(&v, 4); };
    auto put_str = [&](const char* s) { put(s, std::strlen(s) + 1); };
    auto attr    = [&](const char* name, const char* type, const void* data, int32_t size) {
        put_str(name);
        put_str(type);
        put_i32(size);
        put(data, (size_t)size);
    };

    put_i32(20000630); // magic
    put_i32(2);        // version 2, scanline, single part
    {
        std::vector<uint8_t> ch;
        for (const char* c : { "B", "G", "R" }) { // channel list is sorted by name
            ch.insert(ch.end(), c, c + 2);
            const int32_t pixel_type = 2; // FLOAT
            const uint8_t p_linear[4] = { 0, 0, 0, 0 };
            const int32_t sampling[2] = { 1, 1 };
            ch.insert(ch.end(), (const uint8_t*)&pixel_type, (const uint8_t*)&pixel_type + 4);
            ch.insert(ch.end(), p_linear, p_linear + 4);
            ch.insert(ch.end(), (const uint8_t*)sampling, (const uint8_t*)sampling + 8);
        }
        ch.push_back(0);
        attr("channels", "chlist", ch.data(), (int32_t)ch.size());
    }
    const char* want_comp     = std::getenv("IGH_EXR_COMPRESSION");
    const bool zip            = !(want_comp && std::strcmp(want_comp, "none") == 0);
    const uint8_t compression = zip ? 3 : 0; // ZIP_COMPRESSION (16 scanlines per block) : NO_COMPRESSION
    attr("compression", "compression", &compression, 1);
    const int32_t window[4] = { 0, 0, width - 1, height - 1 };
    attr("dataWindow", "box2i", window, 16);
    attr("displayWindow", "box2i", window, 16);
    const uint8_t line_order = 0; // increasing y
    attr("lineOrder", "lineOrder", &line_order, 1);
    const float par = 1.0f;
    attr("pixelAspectRatio", "float", &par, 4);
    const float center[2] = { 0.0f, 0.0f };
    attr("screenWindowCenter", "v2f", center, 8);
    const float sw = 1.0f;
    attr("screenWindowWidth", "float", &sw, 4);
    for (const auto& kv : string_attributes)
        attr(kv.first.c_str(), "string", kv.second.data(), (int32_t)kv.second.size());
    out.push_back(0); // end of header

    const size_t row_bytes   = (size_t)width * 3 * 4;
    const int lines          = zip ? 16 : 1;
    const int blocks         = (height + lines - 1) / lines;
    const size_t table_start = out.size();
    out.resize(table_start + (size_t)blocks * 8);
    std::vector<uint8_t> raw, shuffled, packed;
    for (int b = 0; b < blocks; ++b) {
        const int y0 = b * lines, ny = std::min(lines, height - y0);
        raw.resize(row_bytes * (size_t)ny);
        for (int r = 0; r < ny; ++r) {
            float* dst       = reinterpret_cast<float*>(&raw[(size_t)r * row_bytes]);
            const float* src = rgb + (size_t)(y0 + r) * width * 3;
            for (int c = 0; c < 3; ++c) // B, G, R planes of the scanline
                for (int x = 0; x < width; ++x)
                    dst[(size_t)c * width + x] = src[(size_t)x * 3 + (2 - c)] * scale;
        }
        const uint8_t* payload = raw.data();
        size_t payload_size    = raw.size();
        if (zip) {
            const size_t n = raw.size(), half = (n + 1) / 2;
            shuffled.resize(n);
            for (size_t i = 0; i < n; ++i) // even bytes first, odd bytes second
                shuffled[(i & 1) ? half + i / 2 : i / 2] = raw[i];
            for (size_t i = n - 1; i > 0; --i) // delta predictor, biased by 128
                shuffled[i] = (uint8_t)((int)shuffled[i] - (int)shuffled[i - 1] + 128 + 256);
            uLongf len = compressBound((uLong)n);
            packed.resize(len);
            if (compress2(packed.data(), &len, shuffled.data(), (uLong)n, Z_DEFAULT_COMPRESSION) != Z_OK)
                throw std::runtime_error("EXR '" + path + "': deflate failed");
            if (len < n)
                payload = packed.data(), payload_size = (size_t)len;
        }
        const uint64_t off = out.size();
        std::memcpy(&out[table_start + (size_t)b * 8], &off, 8);
        put_i32(y0);
        put_i32((int32_t)payload_size);
        put(payload, payload_size);
    }
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f)
        throw std::runtime_error("EXR '" + path + "': cannot open for writing");
    const size_t n = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (n != out.size())
        throw std::runtime_error("EXR '" + path + "': short write");
}

} // namespace igh
