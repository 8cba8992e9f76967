// floatimage.h — readers for the two floating-point image formats scenes refer to (Image::load, src/runtime/Image.cpp:497-712;
// the reference delegates to tinyexr and stb_image, neither of which is in this image):
//
//   * OpenEXR, single-part scanline files with HALF / FLOAT / UINT channels, compression NONE, RLE, ZIPS, ZIP and PIZ, written
//     from the published file-layout description ("OpenEXR File Layout"; PIZ = a per-channel 2-D wavelet over 16-bit words, a
//     bitmap-driven lookup table and a canonical Huffman code with a run-length symbol);
//   * Radiance RGBE (.hdr / .pic), flat and new-style run-length encoded scanlines, -Y +X orientation.
//
// Both return rows top to bottom; the texture bank flips them (Image::flipY) like the reference. Channel selection follows
// Image.cpp:593-646: one channel ("Y" or "A") -> gray, otherwise R, G, B (+ A, else 1).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>

namespace igh {

struct FloatImage {
    uint32_t width = 0, height = 0, channels = 0; // channels: 1 or 4
    std::vector<float> pixels;                    // row 0 = top
};

namespace fimg {

[[noreturn]] inline void bad(const std::string& path, const std::string& what) { throw std::runtime_error("Image '" + path + "': " + what); }

inline std::vector<uint8_t> readAll(const std::string& path)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f)
        bad(path, "cannot open");
    std::vector<uint8_t> b;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
        b.insert(b.end(), buf, buf + n);
    std::fclose(f);
    return b;
}

inline float halfToFloat(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h >> 15) << 31;
    const uint32_t e = (h >> 10) & 0x1F, m = h & 0x3FF;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) {
            bits = sign;
        } else { // subnormal: normalise
            int shift = 0;
            uint32_t mm = m;
            while (!(mm & 0x400))
                mm <<= 1, ++shift;
            bits = sign | ((uint32_t)(127 - 15 - shift + 1) << 23) | ((mm & 0x3FF) << 13);
        }
    } else if (e == 31) {
        bits = sign | 0x7F800000u | (m << 13);
    } else {
        bits = sign | ((e + 112) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// ---- the byte-level codecs ---------------------------------------------------------------------------------------------

// ZIP / ZIPS / RLE post-processing: running-sum predictor, then the two halves are the even and the odd bytes
inline void unpredictAndInterleave(std::vector<uint8_t>& t, std::vector<uint8_t>& out)
{
    for (size_t i = 1; i < t.size(); ++i)
        t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
    const size_t n = t.size(), half = (n + 1) / 2;
    out.resize(n);
    for (size_t i = 0; i < n; ++i)
        out[i] = (i & 1) ? t[half + i / 2] : t[i / 2];
}

inline bool inflateTo(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, size_t expect)
{
    dst.resize(expect);
    uLongf len = (uLongf)expect;
    return uncompress(dst.data(), &len, src, (uLong)n) == Z_OK && len == expect;
}

inline bool unRle(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, size_t expect)
{
    dst.clear();
    dst.reserve(expect);
    size_t p = 0;
    while (p < n) {
        const int8_t c = (int8_t)src[p++];
        if (c < 0) { // -c literal bytes
            const size_t k = (size_t)(-(int)c);
            if (p + k > n)
                return false;
            dst.insert(dst.end(), src + p, src + p + k);
            p += k;
        } else { // c + 1 copies of the next byte
            if (p >= n)
                return false;
            dst.insert(dst.end(), (size_t)c + 1, src[p++]);
        }
        if (dst.size() > expect)
            return false;
    }
    return dst.size() == expect;
}

// ---- PIZ ---------------------------------------------------------------------------------------------------------------

struct BitReader { // most significant bit first
    const uint8_t* d;
    size_t n, p = 0;
    uint64_t acc = 0;
    int have    = 0;
    uint64_t consumed = 0;
    BitReader(const uint8_t* data, size_t size) : d(data), n(size) {}
    uint32_t peek(int bits)
    {
        while (have < bits) {
            acc = (acc << 8) | (p < n ? d[p] : 0);
            ++p;
            have += 8;
        }
        return (uint32_t)((acc >> (have - bits)) & ((1ull << bits) - 1));
    }
    void skip(int bits) { have -= bits, consumed += (uint64_t)bits; }
    uint32_t get(int bits)
    {
        const uint32_t v = peek(bits);
        skip(bits);
        return v;
    }
};

// Canonical Huffman over the 16-bit alphabet plus one run-length symbol. The table stores 6-bit code lengths for the symbols
// im..iM (59..62 = short runs of zero lengths, 63 = long run); codes of one length are consecutive in symbol order, the first
// code of a length follows from the counts of the longer ones.
inline bool hufDecode(const uint8_t* data, size_t size, std::vector<uint16_t>& out, size_t n_raw)
{
    if (size < 20)
        return false;
    uint32_t head[5];
    std::memcpy(head, data, 20);
    const uint32_t im = head[0], iM = head[1], n_bits = head[3];
    constexpr uint32_t Alphabet = 65537; // 65536 values + the run-length symbol
    if (im >= Alphabet || iM >= Alphabet)
        return false;
    std::vector<uint8_t> len(Alphabet, 0);
    BitReader tr(data + 20, size - 20);
    for (uint32_t s = im; s <= iM;) {
        const uint32_t l = tr.get(6);
        if (l == 63)
            s += tr.get(8) + 6;
        else if (l >= 59)
            s += l - 59 + 2;
        else
            len[s++] = (uint8_t)l;
    }
    const size_t table_bytes = (size_t)((tr.consumed + 7) / 8);
    uint64_t count[59] = {}, first[59] = {};
    for (uint32_t s = 0; s < Alphabet; ++s)
        ++count[len[s]];
    uint64_t c = 0;
    for (int l = 58; l >= 1; --l) {
        first[l] = c;
        c        = (c + count[l]) >> 1;
    }
    // symbols ordered by (length, symbol) and the index of each length's first symbol in that order
    std::vector<uint32_t> sorted;
    sorted.reserve(Alphabet - count[0]);
    uint64_t base[60] = {};
    {
        uint64_t at = 0;
        for (int l = 1; l <= 58; ++l)
            base[l] = at, at += count[l];
        sorted.resize((size_t)at);
        uint64_t next[59];
        std::memcpy(next, base, sizeof(next));
        for (uint32_t s = 0; s < Alphabet; ++s)
            if (len[s])
                sorted[(size_t)next[len[s]]++] = s;
    }
    int max_len = 0;
    for (int l = 1; l <= 58; ++l)
        if (count[l]) {
            if (l > 56 || first[l] + count[l] > (1ull << l))
                return false; // not a prefix code (corrupt table); codes beyond 56 bits do not occur in real files
            max_len = l;
        }
    // direct table for codes of at most Fast bits
    constexpr int Fast = 12;
    std::vector<int32_t> fsym(1u << Fast, -1);
    std::vector<uint8_t> flen(1u << Fast, 0);
    for (int l = 1; l <= std::min(Fast, max_len); ++l)
        for (uint64_t k = 0; k < count[l]; ++k) {
            const uint64_t code = first[l] + k;
            const uint64_t lo   = code << (Fast - l);
            for (uint64_t j = 0; j < (1ull << (Fast - l)); ++j) {
                fsym[(size_t)(lo + j)] = (int32_t)sorted[(size_t)(base[l] + k)];
                flen[(size_t)(lo + j)] = (uint8_t)l;
            }
        }

    out.resize(n_raw);
    BitReader br(data + 20 + table_bytes, size - 20 - table_bytes);
    size_t o = 0;
    while (o < n_raw) {
        int32_t sym;
        const uint32_t idx = br.peek(Fast);
        if (fsym[idx] >= 0) {
            sym = fsym[idx];
            br.skip(flen[idx]);
        } else {
            sym = -1;
            for (int l = Fast + 1; l <= max_len; ++l) {
                const uint64_t code = br.peek(l);
                if (count[l] && code >= first[l] && code < first[l] + count[l]) {
                    sym = (int32_t)sorted[(size_t)(base[l] + (code - first[l]))];
                    br.skip(l);
                    break;
                }
            }
            if (sym < 0)
                return false;
        }
        if ((uint32_t)sym == iM) {
            const uint32_t rep = br.get(8);
            if (o == 0 || o + rep > n_raw)
                return false;
            for (uint32_t k = 0; k < rep; ++k)
                out[o + k] = out[o - 1];
            o += rep;
        } else {
            out[o++] = (uint16_t)sym;
        }
        if (br.consumed > n_bits)
            return false;
    }
    return true;
}

// inverse of the wavelet's 2 x 1 step: (low, high) -> the two samples; 14-bit data uses the signed form, 16-bit the modular one
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
{
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a            = (uint16_t)(int16_t)ai;
    b            = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
{
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xFFFF;
    const int aa = (d + bb - 0x8000) & 0xFFFF;
    a = (uint16_t)aa, b = (uint16_t)bb;
}

// in place over one 16-bit component: nx samples `ox` apart per row, ny rows `oy` apart
inline void wav2Decode(uint16_t* in, int nx, int ox, int ny, int oy, uint32_t max_value)
{
    const bool w14 = max_value < (1u << 14);
    auto dec       = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { w14 ? wdec14(l, h, a, b) : wdec16(l, h, a, b); };
    const int n    = nx > ny ? ny : nx;
    int p          = 1;
    while (p <= n)
        p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    while (p >= 1) {
        uint16_t* py       = in;
        uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2;
        const ptrdiff_t ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px       = py;
            uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                uint16_t* p10 = px + oy1;
                uint16_t* p11 = p10 + ox1;
                dec(*px, *p10, i00, i10);
                dec(*p01, *p11, i01, i11);
                dec(i00, i01, *px, *p01);
                dec(i10, i11, *p10, *p11);
            }
            if (nx & p) { // a column without a horizontal partner
                uint16_t* p10 = px + oy1;
                dec(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) { // a row without a vertical partner
            uint16_t* px       = py;
            uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                dec(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

inline bool unPiz(const uint8_t* raw, size_t size, const std::vector<int>& words_per_pixel, int nx, int ny, std::vector<uint8_t>& out)
{
    size_t total = 0;
    for (int w : words_per_pixel)
        total += (size_t)w * nx * ny;
    if (size < 4)
        return false;
    uint16_t lo, hi;
    std::memcpy(&lo, raw, 2);
    std::memcpy(&hi, raw + 2, 2);
    size_t pos = 4;
    std::vector<uint8_t> bitmap(8192, 0);
    if (lo <= hi) {
        if (hi >= 8192 || pos + (size_t)(hi - lo + 1) > size)
            return false;
        std::memcpy(&bitmap[lo], raw + pos, (size_t)(hi - lo + 1));
        pos += (size_t)(hi - lo + 1);
    }
    bitmap[0] |= 1; // zero is always present
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t v = 0; v < 65536; ++v)
        if (bitmap[v >> 3] & (1u << (v & 7)))
            lut[k++] = (uint16_t)v;
    const uint32_t max_value = k - 1;
    if (pos + 4 > size)
        return false;
    int32_t length;
    std::memcpy(&length, raw + pos, 4);
    pos += 4;
    if (length < 0 || pos + (size_t)length > size)
        return false;
    std::vector<uint16_t> words;
    if (!hufDecode(raw + pos, (size_t)length, words, total))
        return false;
    // the block holds the channels one after the other, each as [ny][nx][words]; the chunk wants rows of [channel][x][word]
    size_t row_words = 0;
    for (int w : words_per_pixel)
        row_words += (size_t)w * nx;
    out.resize(total * 2);
    uint16_t* dst = reinterpret_cast<uint16_t*>(out.data());
    size_t start = 0, col = 0;
    for (int w : words_per_pixel) {
        uint16_t* block = words.data() + start;
        for (int j = 0; j < w; ++j)
            wav2Decode(block + j, nx, w, ny, w * nx, max_value);
        for (int y = 0; y < ny; ++y)
            for (int i = 0; i < nx * w; ++i)
                dst[(size_t)y * row_words + col + i] = lut[block[(size_t)y * nx * w + i]];
        start += (size_t)w * nx * ny;
        col += (size_t)w * nx;
    }
    return true;
}

} // namespace fimg

// ---- OpenEXR -------------------------------------------------------------------------------------------------------------

inline FloatImage readExr(const std::string& path)
{
    using namespace fimg;
    const std::vector<uint8_t> b = readAll(path);
    if (b.size() < 8)
        bad(path, "not an OpenEXR file");
    uint32_t magic, version;
    std::memcpy(&magic, &b[0], 4);
    std::memcpy(&version, &b[4], 4);
    if (magic != 20000630u)
        bad(path, "not an OpenEXR file");
    if (version & (0x200u | 0x800u | 0x1000u))
        bad(path, "tiled, deep and multi-part OpenEXR files are not supported");
    size_t pos = 8;
    std::map<std::string, std::vector<uint8_t>> attrs;
    auto cstr = [&](size_t& p) {
        const size_t s = p;
        while (p < b.size() && b[p])
            ++p;
        if (p >= b.size())
            bad(path, "truncated header");
        return std::string((const char*)&b[s], p++ - s);
    };
    for (;;) {
        const std::string name = cstr(pos);
        if (name.empty())
            break;
        cstr(pos); // type
        if (pos + 4 > b.size())
            bad(path, "truncated header");
        int32_t n;
        std::memcpy(&n, &b[pos], 4);
        pos += 4;
        if (n < 0 || pos + (size_t)n > b.size())
            bad(path, "truncated header");
        attrs[name].assign(b.begin() + (ptrdiff_t)pos, b.begin() + (ptrdiff_t)(pos + (size_t)n));
        pos += (size_t)n;
    }
    for (const char* need : { "channels", "compression", "dataWindow" })
        if (!attrs.count(need))
            bad(path, std::string("header lacks '") + need + "'");
    struct Chan {
        std::string name;
        int type; // 0 UINT, 1 HALF, 2 FLOAT
    };
    std::vector<Chan> chans;
    {
        const std::vector<uint8_t>& ch = attrs["channels"];
        size_t q = 0;
        while (q < ch.size() && ch[q]) {
            const size_t s = q;
            while (q < ch.size() && ch[q])
                ++q;
            if (q + 17 > ch.size())
                bad(path, "truncated channel list");
            Chan c;
            c.name = std::string((const char*)&ch[s], q - s);
            ++q;
            int32_t v[4];
            std::memcpy(&v[0], &ch[q], 4);
            std::memcpy(&v[2], &ch[q + 8], 8);
            q += 16;
            if (v[0] < 0 || v[0] > 2)
                bad(path, "unknown pixel type");
            if (v[2] != 1 || v[3] != 1)
                bad(path, "subsampled channels are not supported");
            c.type = v[0];
            chans.push_back(c);
        }
    }
    if (chans.empty())
        bad(path, "no channels");
    const int comp = attrs["compression"].empty() ? -1 : attrs["compression"][0];
    int lines;
    switch (comp) {
    case 0: case 1: case 2: lines = 1; break; // NONE, RLE, ZIPS
    case 3: lines = 16; break;                // ZIP
    case 4: lines = 32; break;                // PIZ
    default: bad(path, "compression method " + std::to_string(comp) + " is not supported (NONE, RLE, ZIPS, ZIP, PIZ are)");
    }
    if (attrs["dataWindow"].size() != 16)
        bad(path, "bad dataWindow");
    int32_t win[4];
    std::memcpy(win, attrs["dataWindow"].data(), 16);
    const int64_t W = (int64_t)win[2] - win[0] + 1, H = (int64_t)win[3] - win[1] + 1;
    if (W <= 0 || H <= 0 || W > (1 << 16) || H > (1 << 16))
        bad(path, "bad dataWindow");
    auto px_size = [](int t) { return t == 1 ? 2 : 4; };
    size_t row_bytes = 0;
    std::vector<int> words;
    for (const Chan& c : chans)
        row_bytes += (size_t)px_size(c.type) * (size_t)W, words.push_back(px_size(c.type) / 2);
    const size_t n_chunks = (size_t)((H + lines - 1) / lines);
    if (pos + n_chunks * 8 > b.size())
        bad(path, "truncated offset table");

    // which file channel feeds which output channel (Image.cpp:593-646; names compared in lower case)
    int idx[5] = { -1, -1, -1, -1, -1 }; // R G B A Y
    for (size_t c = 0; c < chans.size(); ++c) {
        std::string n = chans[c].name;
        for (char& x : n)
            x = (char)std::tolower((unsigned char)x);
        if (n.rfind("default.", 0) == 0)
            n = n.substr(8);
        const char* keys = "rgbay";
        if (n.size() == 1)
            for (int k = 0; k < 5; ++k)
                if (n[0] == keys[k])
                    idx[k] = (int)c;
    }
    FloatImage img;
    img.width = (uint32_t)W, img.height = (uint32_t)H;
    int src[4];
    if (chans.size() == 1) {
        img.channels = 1;
        src[0]       = idx[4] != -1 ? idx[4] : idx[3];
        if (src[0] < 0)
            bad(path, "a single channel that is neither 'Y' nor 'A'");
    } else {
        img.channels = 4;
        if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0)
            bad(path, "no R, G and B channels");
        src[0] = idx[0], src[1] = idx[1], src[2] = idx[2], src[3] = idx[3];
    }
    img.pixels.assign((size_t)W * H * img.channels, 1.0f);
    std::vector<size_t> chan_off(chans.size());
    {
        size_t o = 0;
        for (size_t c = 0; c < chans.size(); ++c)
            chan_off[c] = o, o += (size_t)px_size(chans[c].type) * (size_t)W;
    }

    std::vector<uint8_t> tmp, data;
    for (size_t k = 0; k < n_chunks; ++k) {
        uint64_t off;
        std::memcpy(&off, &b[pos + k * 8], 8);
        if (off + 8 > b.size())
            bad(path, "chunk offset outside the file");
        int32_t y, size;
        std::memcpy(&y, &b[off], 4);
        std::memcpy(&size, &b[off + 4], 4);
        if (size < 0 || off + 8 + (uint64_t)size > b.size() || y < win[1] || y > win[3])
            bad(path, "bad chunk");
        const int ny      = (int)std::min<int64_t>(lines, (int64_t)win[3] - y + 1);
        const size_t want = row_bytes * (size_t)ny;
        const uint8_t* raw = &b[off + 8];
        const uint8_t* dec;
        if ((size_t)size == want) { // stored as is when compressing would not have helped
            dec = raw;
        } else if (comp == 2 || comp == 3) {
            if (!inflateTo(raw, (size_t)size, tmp, want))
                bad(path, "corrupt ZIP chunk");
            unpredictAndInterleave(tmp, data);
            dec = data.data();
        } else if (comp == 1) {
            if (!unRle(raw, (size_t)size, tmp, want))
                bad(path, "corrupt RLE chunk");
            unpredictAndInterleave(tmp, data);
            dec = data.data();
        } else if (comp == 4) {
            if (!unPiz(raw, (size_t)size, words, (int)W, ny, data) || data.size() != want)
                bad(path, "corrupt PIZ chunk");
            dec = data.data();
        } else {
            bad(path, "chunk size does not match an uncompressed image");
        }
        for (int r = 0; r < ny; ++r) {
            const uint8_t* row = dec + (size_t)r * row_bytes;
            float* dst         = &img.pixels[(size_t)(y - win[1] + r) * (size_t)W * img.channels];
            for (uint32_t oc = 0; oc < img.channels; ++oc) {
                const int c = src[oc];
                if (c < 0)
                    continue; // no alpha channel: stays 1
                const uint8_t* p = row + chan_off[(size_t)c];
                for (int64_t x = 0; x < W; ++x) {
                    float v;
                    if (chans[(size_t)c].type == 1) {
                        uint16_t h;
                        std::memcpy(&h, p + x * 2, 2);
                        v = halfToFloat(h);
                    } else if (chans[(size_t)c].type == 2) {
                        std::memcpy(&v, p + x * 4, 4);
                    } else {
                        uint32_t u;
                        std::memcpy(&u, p + x * 4, 4);
                        v = (float)u;
                    }
                    dst[(size_t)x * img.channels + oc] = v;
                }
            }
        }
    }
    return img;
}

// ---- Radiance RGBE -------------------------------------------------------------------------------------------------------

inline FloatImage readHdr(const std::string& path)
{
    using namespace fimg;
    const std::vector<uint8_t> b = readAll(path);
    size_t pos = 0;
    auto line  = [&]() {
        std::string s;
        while (pos < b.size() && b[pos] != '\n')
            s.push_back((char)b[pos++]);
        ++pos;
        return s;
    };
    const std::string sig = line();
    if (sig != "#?RADIANCE" && sig != "#?RGBE")
        bad(path, "not a Radiance picture");
    bool format_ok = false;
    for (;;) {
        if (pos >= b.size())
            bad(path, "truncated header");
        const std::string s = line();
        if (s.empty())
            break;
        if (s == "FORMAT=32-bit_rle_rgbe")
            format_ok = true;
    }
    if (!format_ok)
        bad(path, "only FORMAT=32-bit_rle_rgbe is supported");
    const std::string res = line();
    int H = 0, W = 0;
    if (std::sscanf(res.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0 || W > (1 << 16) || H > (1 << 16))
        bad(path, "only the -Y h +X w orientation is supported");
    FloatImage img;
    img.width = (uint32_t)W, img.height = (uint32_t)H, img.channels = 4;
    img.pixels.resize((size_t)W * H * 4);
    // mantissa * 2^(e - 136); a zero exponent byte is black (stb_image's conversion, which is what the reference loads through)
    auto convert = [](const uint8_t* rgbe, float* out) {
        if (rgbe[3] != 0) {
            const float f = std::ldexp(1.0f, (int)rgbe[3] - (128 + 8));
            out[0] = rgbe[0] * f, out[1] = rgbe[1] * f, out[2] = rgbe[2] * f;
        } else {
            out[0] = out[1] = out[2] = 0;
        }
        out[3] = 1;
    };
    std::vector<uint8_t> scan((size_t)W * 4);
    for (int y = 0; y < H; ++y) {
        if (pos + 4 > b.size())
            bad(path, "truncated pixel data");
        const bool rle = W >= 8 && W < 32768 && b[pos] == 2 && b[pos + 1] == 2 && !(b[pos + 2] & 0x80);
        if (!rle) { // flat scanline (old-style run markers are not used by any current writer)
            if (pos + (size_t)W * 4 > b.size())
                bad(path, "truncated pixel data");
            std::memcpy(scan.data(), &b[pos], (size_t)W * 4);
            pos += (size_t)W * 4;
            for (int x = 0; x < W; ++x)
                convert(&scan[(size_t)x * 4], &img.pixels[((size_t)y * W + x) * 4]);
            continue;
        }
        if ((((int)b[pos + 2]) << 8 | b[pos + 3]) != W)
            bad(path, "scanline length does not match the image width");
        pos += 4;
        for (int c = 0; c < 4; ++c) { // the four components one after the other
            int x = 0;
            while (x < W) {
                if (pos >= b.size())
                    bad(path, "truncated pixel data");
                int count = b[pos++];
                if (count > 128) { // run
                    count -= 128;
                    if (pos >= b.size() || x + count > W)
                        bad(path, "corrupt run");
                    const uint8_t v = b[pos++];
                    for (int k = 0; k < count; ++k)
                        scan[(size_t)(x++) * 4 + c] = v;
                } else { // literals
                    if (count == 0 || pos + (size_t)count > b.size() || x + count > W)
                        bad(path, "corrupt scanline");
                    for (int k = 0; k < count; ++k)
                        scan[(size_t)(x++) * 4 + c] = b[pos++];
                }
            }
        }
        for (int x = 0; x < W; ++x)
            convert(&scan[(size_t)x * 4], &img.pixels[((size_t)y * W + x) * 4]);
    }
    return img;
}

inline bool isFloatImagePath(const std::string& path)
{
    auto ends = [&](const char* e) {
        const size_t n = std::strlen(e);
        if (path.size() < n)
            return false;
        for (size_t i = 0; i < n; ++i)
            if (std::tolower((unsigned char)path[path.size() - n + i]) != e[i])
                return false;
        return true;
    };
    return ends(".exr") || ends(".hdr");
}

inline FloatImage readFloatImage(const std::string& path)
{
    const size_t n = path.size();
    if (n >= 4 && std::tolower((unsigned char)path[n - 3]) == 'h')
        return readHdr(path);
    return readExr(path);
}

} // namespace igh
