// png.h — minimal PNG reader for the loader's bitmap textures (zlib for the inflate step).
//
// The reference loads 8-bit images through stb_image (src/runtime/Image.cpp:714-808, `Image::loadAsPacked`)
// with vertical flip; this covers what that path yields for PNG files: every colour type (gray, gray+alpha, RGB, RGBA,
// palette with optional tRNS alpha) at every bit depth the format allows, reduced to 8 bits per channel as stb_image
// does. Interlaced files are refused with a clear message; header dimensions are bounded before any size arithmetic.
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
    std::vector<uint8_t> idat, palette, trns;
    bool have_header = false;
    int depth = 0, ctype = 0;
    uint32_t file_channels = 0;
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
            depth = body[8], ctype = body[9];
            const int interlace = body[12];
            // the header is untrusted: bound the dimensions before any size arithmetic (stb_image's limit is 2^24 per side;
            // textures of this loader are far below 65536)
            if (img.width == 0 || img.height == 0 || img.width > 65536u || img.height > 65536u)
                throw bad("image dimensions out of range (1 .. 65536)");
            if (interlace != 0)
                throw bad("interlaced PNG files are not supported by this loader");
            const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                                : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                             : (depth == 8 || depth == 16);
            if (!depth_ok)
                throw bad("bit depth not allowed for this colour type");
            switch (ctype) {
            case 0: file_channels = 1; break;
            case 2: file_channels = 3; break;
            case 3: file_channels = 1; break; // palette index
            case 4: file_channels = 2; break;
            case 6: file_channels = 4; break;
            default: throw bad("unknown colour type");
            }
            have_header = true;
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768)
                throw bad("bad PLTE chunk");
            palette.assign(body, body + len);
        } else if (std::memcmp(type, "tRNS", 4) == 0) {
            trns.assign(body, body + len);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), body, body + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_header)
        throw bad("missing header");
    if (ctype == 3 && palette.empty())
        throw bad("palette image without a PLTE chunk");

    // bytes per complete pixel for the filters (at least 1), bytes per row as stored
    const size_t bits_pp = (size_t)file_channels * (size_t)depth;
    const size_t bpp     = bits_pp >= 8 ? bits_pp / 8 : 1;
    const size_t stride  = ((size_t)img.width * bits_pp + 7) / 8; // <= 65536 * 8
    const size_t raw_size = (stride + 1) * (size_t)img.height;    // <= 2^35: no overflow in size_t
    if (raw_size > ((size_t)1 << 32))
        throw bad("image too large");
    std::vector<uint8_t> raw(raw_size);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        throw bad("corrupt image data");

    // undo the per-row filters (PNG specification, section 9), in place
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < img.height; ++y) {
        const uint8_t ft  = raw[(stride + 1) * y];
        uint8_t* out      = &raw[(stride + 1) * y + 1];
        const uint8_t* up = y ? &raw[(stride + 1) * (y - 1) + 1] : zero.data();
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
            out[x] = (uint8_t)(out[x] + pred);
        }
    }

    // expand to 8 bits per channel the way stb_image does: 16-bit samples keep their high byte, 1/2/4-bit gray is scaled to
    // 0 .. 255, palette indices become RGB (RGBA with a tRNS chunk)
    const bool pal_alpha = ctype == 3 && !trns.empty();
    img.channels         = ctype == 3 ? (pal_alpha ? 4u : 3u) : file_channels;
    img.data.resize((size_t)img.width * img.height * img.channels);
    const int gray_scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    for (uint32_t y = 0; y < img.height; ++y) {
        const uint8_t* row = &raw[(stride + 1) * y + 1];
        uint8_t* out       = &img.data[(size_t)y * img.width * img.channels];
        for (uint32_t x = 0; x < img.width; ++x) {
            if (depth == 16) {
                for (uint32_t c = 0; c < file_channels; ++c)
                    out[(size_t)x * file_channels + c] = row[((size_t)x * file_channels + c) * 2];
            } else if (depth == 8 && ctype != 3) {
                for (uint32_t c = 0; c < file_channels; ++c)
                    out[(size_t)x * file_channels + c] = row[(size_t)x * file_channels + c];
            } else {
                // packed samples, most significant bits first (depth 8 for palette indices falls through here too)
                const size_t bit = (size_t)x * (size_t)depth;
                const int v      = (row[bit / 8] >> (8 - depth - (int)(bit % 8))) & ((1 << depth) - 1);
                if (ctype == 3) {
                    if ((size_t)v * 3 + 2 >= palette.size())
                        throw bad("palette index out of range");
                    out[(size_t)x * img.channels + 0] = palette[(size_t)v * 3 + 0];
                    out[(size_t)x * img.channels + 1] = palette[(size_t)v * 3 + 1];
                    out[(size_t)x * img.channels + 2] = palette[(size_t)v * 3 + 2];
                    if (pal_alpha)
                        out[(size_t)x * 4 + 3] = (size_t)v < trns.size() ? trns[v] : 255;
                } else {
                    out[x] = (uint8_t)(v * gray_scale);
                }
            }
        }
    }
    return img;
}

} // namespace igh
