// png.h — minimal PNG reader for the loader's bitmap textures (zlib for the inflate step).
//
// The reference loads 8-bit images through stb_image (src/runtime/Image.cpp:714-808, `Image::loadAsPacked`)
// with vertical flip; this covers what that path yields for ordinary PNG files: bit depth 8, colour types
// gray / gray+alpha / RGB / RGBA, no interlacing. Anything else is refused with a clear message.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace igh {

struct PngImage {
    uint32_t width = 0, height = 0;
    uint32_t channels = 0;     // as stored in the file: 1, 2, 3 or 4
    std::vector<uint8_t> data; // rows top to bottom, `channels` bytes per pixel
};

inline PngImage readPng(const std::string& path)
{
    auto bad = [&](const std::string& why) -> std::runtime_error { return std::runtime_error("PNG '" + path + "': " + why); };

    std::vector<uint8_t> file;
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f)
            throw bad("cannot open file");
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        file.resize(n > 0 ? (size_t)n : 0);
        const size_t got = file.empty() ? 0 : std::fread(file.data(), 1, file.size(), f);
        std::fclose(f);
        if (got != file.size())
            throw bad("short read");
    }
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0)
        throw bad("not a PNG file");

    auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | ((uint32_t)file[o + 1] << 16) | ((uint32_t)file[o + 2] << 8) | (uint32_t)file[o + 3]; };

    PngImage img;
    std::vector<uint8_t> idat;
    bool have_header = false;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(pos);
        if (pos + 12 + (size_t)len > file.size())
            throw bad("truncated chunk");
        const char* type    = reinterpret_cast<const char*>(&file[pos + 4]);
        const uint8_t* body = &file[pos + 8];
        if (std::memcmp(type, "IHDR", 4) == 0) {
            if (len != 13)
                throw bad("bad IHDR");
            img.width  = be32(pos + 8);
            img.height = be32(pos + 12);
            const int depth = body[8], ctype = body[9], interlace = body[12];
            if (depth != 8)
                throw bad("only 8-bit PNG files are supported by this loader");
            if (interlace != 0)
                throw bad("interlaced PNG files are not supported by this loader");
            switch (ctype) {
            case 0: img.channels = 1; break;
            case 2: img.channels = 3; break;
            case 4: img.channels = 2; break;
            case 6: img.channels = 4; break;
            default: throw bad("palette PNG files are not supported by this loader");
            }
            have_header = true;
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), body, body + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_header || img.width == 0 || img.height == 0)
        throw bad("missing header");

    const size_t bpp    = img.channels;
    const size_t stride = (size_t)img.width * bpp;
    std::vector<uint8_t> raw((stride + 1) * img.height);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        throw bad("corrupt image data");

    // undo the per-row filters (PNG specification, section 9)
    img.data.resize(stride * img.height);
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < img.height; ++y) {
        const uint8_t ft   = raw[(stride + 1) * y];
        const uint8_t* in  = &raw[(stride + 1) * y + 1];
        uint8_t* out       = &img.data[stride * y];
        const uint8_t* up  = y ? &img.data[stride * (y - 1)] : zero.data();
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? out[x - bpp] : 0;
            const int b = up[x];
            const int c = x >= bpp ? up[x - bpp] : 0;
            int pred    = 0;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) / 2; break;
            case 4: {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred        = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: throw bad("unknown row filter");
            }
            out[x] = (uint8_t)(in[x] + pred);
        }
    }
    return img;
}

} // namespace igh
