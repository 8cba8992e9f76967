// exr.h — minimal OpenEXR writer for Runtime::saveFramebuffer (src/runtime/Runtime.cpp:794-876, Image::save):
// single-part scanline file, 32-bit float channels, ZIP-less (NO_COMPRESSION), channels "B", "G", "R" like the
// reference's swizzle. Every OpenEXR reader accepts this form; the reference compresses, which changes bytes on
// disk, not pixel values.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

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
    const uint8_t compression = 0;
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
    const size_t table_start = out.size();
    const size_t data_start  = table_start + (size_t)height * 8;
    out.resize(data_start + (size_t)height * (8 + row_bytes));
    for (int y = 0; y < height; ++y) {
        const uint64_t off = data_start + (size_t)y * (8 + row_bytes);
        std::memcpy(&out[table_start + (size_t)y * 8], &off, 8);
        const int32_t yy = y, sz = (int32_t)row_bytes;
        std::memcpy(&out[off], &yy, 4);
        std::memcpy(&out[off + 4], &sz, 4);
        float* dst = reinterpret_cast<float*>(&out[off + 8]);
        const float* src = rgb + (size_t)y * width * 3;
        for (int c = 0; c < 3; ++c)      // B, G, R planes of the scanline
            for (int x = 0; x < width; ++x)
                dst[(size_t)c * width + x] = src[(size_t)x * 3 + (2 - c)] * scale;
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
