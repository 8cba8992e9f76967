// jpeg.h — JPEG reader for the loader's bitmap textures (Image::loadAsPacked, src/runtime/Image.cpp:714-808, reads them through
// stb_image, which is not in this image). Written from ITU-T T.81: baseline and extended sequential (SOF0 / SOF1) and
// progressive (SOF2) Huffman-coded files, 8 bits per sample, one or three components (gray, YCbCr per JFIF), any sampling
// factors, restart intervals. Chroma is brought to full resolution with the triangle filter stb_image and libjpeg use for 2:1
// factors (nearest otherwise); the inverse DCT is evaluated in floating point, so single samples can differ by one level from a
// fixed-point implementation. Not handled (an error says so): arithmetic coding, lossless and hierarchical modes, 12-bit data,
// CMYK / Adobe transforms.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace igh {

struct JpegImage {
    uint32_t width = 0, height = 0, channels = 0; // 1 or 3
    std::vector<uint8_t> data;                    // rows top to bottom
};

namespace jpg {

[[noreturn]] inline void bad(const std::string& path, const std::string& what) { throw std::runtime_error("JPEG '" + path + "': " + what); }

static const uint8_t kZigZag[64] = { 0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct Huffman {
    bool present = false;
    // canonical code: for every length the first code, the index of its first symbol and the number of codes
    int32_t first_code[17] = {}, first_sym[17] = {}, count[17] = {};
    uint8_t symbols[256] = {};
    // direct table for codes of at most 9 bits: (length << 8) | symbol, 0 = longer code
    uint16_t fast[512] = {};
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;      // Huffman table selectors of the current scan
    int blocks_w = 0, blocks_h = 0; // blocks in the (padded) component plane
    int dc_pred = 0;
    std::vector<int16_t> coef; // blocks_w * blocks_h * 64, natural (row-major) order inside a block
    std::vector<uint8_t> plane; // blocks_w * 8 by blocks_h * 8 samples after the inverse transform
};

struct Decoder {
    const std::string& path;
    const std::vector<uint8_t>& b;
    size_t pos = 0;
    // entropy-coded segment reader
    uint32_t bit_buf = 0;
    int bit_cnt      = 0;
    bool hit_marker  = false;

    uint16_t quant[4][64] = {};
    bool quant_present[4] = {};
    Huffman dc_tab[4], ac_tab[4];
    std::vector<Component> comps;
    int width = 0, height = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool progressive = false, have_frame = false;
    int restart_interval = 0;
    int eob_run          = 0;
    int adobe_transform  = -1;

    Decoder(const std::string& p, const std::vector<uint8_t>& bytes) : path(p), b(bytes) {}

    uint8_t u8()
    {
        if (pos >= b.size())
            bad(path, "truncated file");
        return b[pos++];
    }
    uint16_t u16()
    {
        const uint16_t hi = u8();
        return (uint16_t)((hi << 8) | u8());
    }

    // ---- bits of an entropy-coded segment (0xFF00 = a literal 0xFF; any other marker ends the segment: zeros follow)
    void fill()
    {
        while (bit_cnt <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && pos < b.size()) {
                byte = b[pos];
                if (byte == 0xFF) {
                    const uint8_t next = pos + 1 < b.size() ? b[pos + 1] : 0xD9;
                    if (next == 0x00) {
                        pos += 2;
                    } else {
                        hit_marker = true;
                        byte       = 0;
                    }
                } else {
                    ++pos;
                }
            }
            bit_buf |= byte << (24 - bit_cnt);
            bit_cnt += 8;
        }
    }
    int bits(int n)
    {
        if (n == 0)
            return 0;
        fill();
        const int v = (int)(bit_buf >> (32 - n));
        bit_buf <<= n;
        bit_cnt -= n;
        return v;
    }
    int bit() { return bits(1); }
    // EXTEND of T.81 F.2.2.1
    int receive_extend(int s)
    {
        if (s == 0)
            return 0;
        const int v = bits(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    int decode(const Huffman& h)
    {
        if (!h.present)
            bad(path, "scan uses a Huffman table that was never defined");
        fill();
        const uint16_t f = h.fast[bit_buf >> 23];
        if (f) {
            const int len = f >> 8;
            bit_buf <<= len;
            bit_cnt -= len;
            return f & 0xFF;
        }
        int code = (int)(bit_buf >> 22); // the first 10 bits
        for (int len = 10; len <= 16; ++len) {
            if (h.count[len] && code >= h.first_code[len] && code < h.first_code[len] + h.count[len]) {
                bit_buf <<= len;
                bit_cnt -= len;
                return h.symbols[h.first_sym[len] + code - h.first_code[len]];
            }
            code = (int)(bit_buf >> (21 - (len - 10))); // one more bit
        }
        bad(path, "corrupt entropy-coded data");
    }
    void reset_bits()
    {
        bit_buf = 0, bit_cnt = 0, hit_marker = false;
    }

    // ---- tables
    void read_dqt(size_t end)
    {
        while (pos < end) {
            const uint8_t pq_tq = u8();
            const int pq = pq_tq >> 4, tq = pq_tq & 15;
            if (tq > 3 || pq > 1)
                bad(path, "bad quantisation table");
            for (int i = 0; i < 64; ++i)
                quant[tq][kZigZag[i]] = pq ? u16() : u8();
            quant_present[tq] = true;
        }
    }
    void read_dht(size_t end)
    {
        while (pos < end) {
            const uint8_t tc_th = u8();
            const int tc = tc_th >> 4, th = tc_th & 15;
            if (tc > 1 || th > 3)
                bad(path, "bad Huffman table");
            Huffman& h = tc ? ac_tab[th] : dc_tab[th];
            h          = Huffman{};
            int total  = 0;
            for (int len = 1; len <= 16; ++len) {
                h.count[len] = u8();
                total += h.count[len];
            }
            if (total > 256)
                bad(path, "bad Huffman table");
            for (int i = 0; i < total; ++i)
                h.symbols[i] = u8();
            int code = 0, sym = 0;
            for (int len = 1; len <= 16; ++len) {
                h.first_code[len] = code;
                h.first_sym[len]  = sym;
                if (code + h.count[len] > (1 << len))
                    bad(path, "Huffman table is not a prefix code");
                if (len <= 9)
                    for (int k = 0; k < h.count[len]; ++k) {
                        const int c = (code + k) << (9 - len);
                        for (int j = 0; j < (1 << (9 - len)); ++j)
                            h.fast[c + j] = (uint16_t)((len << 8) | h.symbols[sym + k]);
                    }
                code = (code + h.count[len]) << 1;
                sym += h.count[len];
            }
            h.present = true;
        }
    }
    void read_sof(int marker, size_t end)
    {
        if (have_frame)
            bad(path, "more than one frame");
        progressive = marker == 0xC2;
        if (u8() != 8)
            bad(path, "only 8-bit samples are supported");
        height = u16();
        width  = u16();
        const int n = u8();
        if (width <= 0 || height <= 0 || width > (1 << 15) || height > (1 << 15))
            bad(path, "unsupported image size");
        if (n != 1 && n != 3)
            bad(path, "only gray and YCbCr images are supported (" + std::to_string(n) + " components)");
        comps.resize((size_t)n);
        for (Component& c : comps) {
            c.id = u8();
            const uint8_t hv = u8();
            c.h = hv >> 4, c.v = hv & 15;
            c.tq = u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                bad(path, "bad component description");
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        if (pos != end)
            bad(path, "bad frame header");
        mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
        mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (Component& c : comps) {
            c.blocks_w = mcus_x * c.h;
            c.blocks_h = mcus_y * c.v;
            c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        have_frame = true;
    }

    // ---- one block of a scan
    void block_baseline(Component& c, int16_t* blk)
    {
        const int t = decode(dc_tab[c.td]);
        if (t > 15)
            bad(path, "corrupt DC coefficient");
        c.dc_pred += receive_extend(t);
        blk[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64;) {
            const int rs = decode(ac_tab[c.ta]);
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15)
                    break; // end of block
                k += 16;
                continue;
            }
            k += r;
            if (k > 63)
                bad(path, "corrupt AC coefficients");
            blk[kZigZag[k++]] = (int16_t)receive_extend(s);
        }
    }
    void block_dc_progressive(Component& c, int16_t* blk, int ah, int al)
    {
        if (ah == 0) {
            const int t = decode(dc_tab[c.td]);
            if (t > 15)
                bad(path, "corrupt DC coefficient");
            c.dc_pred += receive_extend(t);
            blk[0] = (int16_t)(c.dc_pred * (1 << al));
        } else if (bit()) {
            blk[0] = (int16_t)(blk[0] | (1 << al));
        }
    }
    void block_ac_first(Component& c, int16_t* blk, int ss, int se, int al)
    {
        if (eob_run > 0) {
            --eob_run;
            return;
        }
        for (int k = ss; k <= se;) {
            const int rs = decode(ac_tab[c.ta]);
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r)
                        eob_run += bits(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > 63)
                bad(path, "corrupt AC coefficients");
            blk[kZigZag[k++]] = (int16_t)(receive_extend(s) * (1 << al));
        }
    }
    void block_ac_refine(Component& c, int16_t* blk, int ss, int se, int al)
    {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k        = ss;
        if (eob_run == 0) {
            for (; k <= se;) {
                const int rs = decode(ac_tab[c.ta]);
                int r        = rs >> 4;
                const int s  = rs & 15;
                int value    = 0;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r);
                        if (r)
                            eob_run += bits(r);
                        break; // the rest of the band only receives correction bits
                    }
                    // r == 15: skip 16 zero-history coefficients
                } else {
                    if (s != 1)
                        bad(path, "corrupt refinement scan");
                    value = bit() ? p1 : m1;
                }
                while (k <= se) {
                    int16_t& coef = blk[kZigZag[k++]];
                    if (coef != 0) {
                        if (bit() && (coef & p1) == 0)
                            coef = (int16_t)(coef + (coef >= 0 ? p1 : m1));
                    } else {
                        if (r == 0) {
                            if (value)
                                coef = (int16_t)value;
                            break;
                        }
                        --r;
                    }
                }
            }
        }
        if (eob_run > 0) {
            for (; k <= se; ++k) {
                int16_t& coef = blk[kZigZag[k]];
                if (coef != 0 && bit() && (coef & p1) == 0)
                    coef = (int16_t)(coef + (coef >= 0 ? p1 : m1));
            }
            --eob_run;
        }
    }

    void read_scan(size_t end)
    {
        if (!have_frame)
            bad(path, "scan before the frame header");
        const int ns = u8();
        if (ns < 1 || ns > (int)comps.size())
            bad(path, "bad scan header");
        std::vector<Component*> sc;
        for (int i = 0; i < ns; ++i) {
            const int id = u8();
            const uint8_t t = u8();
            Component* found = nullptr;
            for (Component& c : comps)
                if (c.id == id)
                    found = &c;
            if (!found)
                bad(path, "scan names an unknown component");
            found->td = t >> 4, found->ta = t & 15;
            if (found->td > 3 || found->ta > 3)
                bad(path, "bad scan header");
            sc.push_back(found);
        }
        const int ss = u8(), se = u8();
        const uint8_t a = u8();
        const int ah = a >> 4, al = a & 15;
        if (pos != end)
            bad(path, "bad scan header");
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13 || ah > 13)
                bad(path, "bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
            bad(path, "bad sequential scan parameters");
        }
        reset_bits();
        for (Component* c : sc)
            c->dc_pred = 0;
        eob_run = 0;

        auto one_block = [&](Component& c, int bx, int by) {
            int16_t* blk = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
            if (!progressive)
                block_baseline(c, blk);
            else if (ss == 0)
                block_dc_progressive(c, blk, ah, al);
            else if (ah == 0)
                block_ac_first(c, blk, ss, se, al);
            else
                block_ac_refine(c, blk, ss, se, al);
        };
        int until_restart = restart_interval;
        int expected_rst  = 0;
        auto restart      = [&]() {
            if (restart_interval == 0 || --until_restart > 0)
                return;
            // byte-align, expect RSTn
            reset_bits();
            while (pos + 1 < b.size() && !(b[pos] == 0xFF && b[pos + 1] >= 0xD0 && b[pos + 1] <= 0xD7)) {
                if (b[pos] == 0xFF && b[pos + 1] != 0 && b[pos + 1] != 0xFF)
                    return; // another marker: the scan ends here
                ++pos;
            }
            if (pos + 1 >= b.size())
                return;
            if (b[pos + 1] != 0xD0 + expected_rst)
                bad(path, "restart markers out of order");
            pos += 2;
            expected_rst  = (expected_rst + 1) & 7;
            until_restart = restart_interval;
            for (Component* c : sc)
                c->dc_pred = 0;
            eob_run = 0;
        };
        if (ns == 1) {
            // non-interleaved: the component's own blocks, only those that cover the image
            Component& c  = *sc[0];
            const int bw = (((width * c.h + hmax - 1) / hmax) + 7) / 8, bh = (((height * c.v + vmax - 1) / vmax) + 7) / 8;
            const int total = bw * bh;
            for (int i = 0; i < total; ++i) {
                one_block(c, i % bw, i / bw);
                if (i + 1 < total)
                    restart();
            }
        } else {
            const int total = mcus_x * mcus_y;
            for (int m = 0; m < total; ++m) {
                const int mx = m % mcus_x, my = m / mcus_x;
                for (Component* c : sc)
                    for (int v = 0; v < c->v; ++v)
                        for (int h = 0; h < c->h; ++h)
                            one_block(*c, mx * c->h + h, my * c->v + v);
                if (m + 1 < total)
                    restart();
            }
        }
        // leave the entropy-coded segment: continue at the next marker
        if (!hit_marker) {
            while (pos + 1 < b.size() && !(b[pos] == 0xFF && b[pos + 1] != 0 && !(b[pos + 1] >= 0xD0 && b[pos + 1] <= 0xD7)))
                ++pos;
        }
        reset_bits();
    }

    // ---- reconstruction
    static void idct8x8(const float* in, float* out)
    {
        // separable, straight from the definition (T.81 A.3.3): out[y][x] = 1/4 sum C(u) C(v) in[v][u] cos cos
        static float basis[8][8];
        static bool ready = false;
        if (!ready) {
            for (int x = 0; x < 8; ++x)
                for (int u = 0; u < 8; ++u)
                    basis[x][u] = (float)((u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16) * 0.5);
            ready = true;
        }
        float tmp[64];
        for (int v = 0; v < 8; ++v)
            for (int x = 0; x < 8; ++x) {
                float s = 0;
                for (int u = 0; u < 8; ++u)
                    s += basis[x][u] * in[v * 8 + u];
                tmp[v * 8 + x] = s;
            }
        for (int x = 0; x < 8; ++x)
            for (int y = 0; y < 8; ++y) {
                float s = 0;
                for (int v = 0; v < 8; ++v)
                    s += basis[y][v] * tmp[v * 8 + x];
                out[y * 8 + x] = s;
            }
    }
    void reconstruct()
    {
        for (Component& c : comps) {
            if (!quant_present[c.tq])
                bad(path, "component uses a quantisation table that was never defined");
            const int pw = c.blocks_w * 8;
            c.plane.assign((size_t)pw * c.blocks_h * 8, 0);
            float in[64], out[64];
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx) {
                    const int16_t* blk = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
                    for (int i = 0; i < 64; ++i)
                        in[i] = (float)blk[i] * (float)quant[c.tq][i];
                    idct8x8(in, out);
                    for (int y = 0; y < 8; ++y)
                        for (int x = 0; x < 8; ++x) {
                            const float v = std::floor(out[y * 8 + x] + 128.5f);
                            c.plane[(size_t)(by * 8 + y) * pw + bx * 8 + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                        }
                }
        }
    }
    // a component plane at full resolution: `fw` x `fh` samples
    std::vector<uint8_t> upsample(const Component& c) const
    {
        const int sx = hmax / c.h, sy = vmax / c.v;
        const int pw = c.blocks_w * 8;
        const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax; // valid samples of the plane
        std::vector<uint8_t> full((size_t)width * height);
        if (hmax % c.h || vmax % c.v)
            bad(path, "fractional sampling factors are not supported");
        if (sx == 1 && sy == 1) {
            for (int y = 0; y < height; ++y)
                std::memcpy(&full[(size_t)y * width], &c.plane[(size_t)y * pw], (size_t)width);
            return full;
        }
        const bool tri_x = sx == 2, tri_y = sy == 2;
        // vertical pass into rows of the component's width, then horizontal; triangle weights 3/4, 1/4 for a factor of two
        std::vector<int> rowbuf((size_t)cw);
        for (int y = 0; y < height; ++y) {
            if (tri_y) {
                const int near = y / 2, far_ = (y & 1) ? (near + 1 < ch ? near + 1 : near) : (near > 0 ? near - 1 : near);
                for (int x = 0; x < cw; ++x)
                    rowbuf[(size_t)x] = 3 * c.plane[(size_t)near * pw + x] + c.plane[(size_t)far_ * pw + x]; // scaled by 4
            } else {
                const int src = y / sy < ch ? y / sy : ch - 1;
                for (int x = 0; x < cw; ++x)
                    rowbuf[(size_t)x] = 4 * c.plane[(size_t)src * pw + x];
            }
            uint8_t* dst = &full[(size_t)y * width];
            for (int x = 0; x < width; ++x) {
                int v16; // scaled by 16
                if (tri_x) {
                    const int near = x / 2, far_ = (x & 1) ? (near + 1 < cw ? near + 1 : near) : (near > 0 ? near - 1 : near);
                    v16 = 3 * rowbuf[(size_t)near] + rowbuf[(size_t)far_];
                } else {
                    const int src = x / sx < cw ? x / sx : cw - 1;
                    v16 = 4 * rowbuf[(size_t)src];
                }
                dst[x] = (uint8_t)((v16 + 8) >> 4);
            }
        }
        return full;
    }

    JpegImage run()
    {
        if (b.size() < 4 || b[0] != 0xFF || b[1] != 0xD8)
            bad(path, "not a JPEG file");
        pos = 2;
        bool done = false;
        while (!done) {
            // next marker (fill bytes 0xFF may precede it)
            uint8_t m = u8();
            if (m != 0xFF)
                bad(path, "expected a marker");
            do {
                m = u8();
            } while (m == 0xFF);
            if (m == 0xD9) {
                done = true;
                break;
            }
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7))
                continue; // stand-alone markers
            const size_t len = u16();
            if (len < 2 || pos + len - 2 > b.size())
                bad(path, "truncated segment");
            const size_t end = pos + len - 2;
            switch (m) {
            case 0xDB: read_dqt(end); break;
            case 0xC4: read_dht(end); break;
            case 0xC0: case 0xC1: case 0xC2: read_sof(m, end); break;
            case 0xDD: restart_interval = u16(); break;
            case 0xDA:
                read_scan(end);
                continue; // read_scan leaves `pos` at the next marker
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                bad(path, "lossless, hierarchical and arithmetic-coded JPEG files are not supported");
            case 0xEE: // Adobe: transform 0 = the components are R, G, B as they are, 1 = YCbCr, 2 = YCCK (not handled)
                if (len >= 14 && std::memcmp(&b[pos], "Adobe", 5) == 0)
                    adobe_transform = b[pos + 11];
                break;
            default: break; // APPn, COM, ...
            }
            pos = end;
            if (pos >= b.size())
                break;
        }
        if (!have_frame)
            bad(path, "no frame");
        reconstruct();
        JpegImage img;
        img.width = (uint32_t)width, img.height = (uint32_t)height, img.channels = (uint32_t)comps.size();
        img.data.resize((size_t)width * height * img.channels);
        if (comps.size() == 1) {
            const std::vector<uint8_t> y = upsample(comps[0]);
            img.data                     = y;
            return img;
        }
        if (adobe_transform == 2)
            bad(path, "YCCK / CMYK files are not supported");
        const std::vector<uint8_t> Y = upsample(comps[0]), Cb = upsample(comps[1]), Cr = upsample(comps[2]);
        if (adobe_transform == 0) { // already R, G, B
            for (size_t i = 0; i < (size_t)width * height; ++i)
                img.data[i * 3 + 0] = Y[i], img.data[i * 3 + 1] = Cb[i], img.data[i * 3 + 2] = Cr[i];
            return img;
        }
        auto clamp8 = [](float v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : (int)(v + 0.5f))); };
        for (size_t i = 0; i < (size_t)width * height; ++i) {
            const float y = Y[i], cb = (float)Cb[i] - 128, cr = (float)Cr[i] - 128; // JFIF 1.02
            img.data[i * 3 + 0] = clamp8(y + 1.402f * cr);
            img.data[i * 3 + 1] = clamp8(y - 0.344136f * cb - 0.714136f * cr);
            img.data[i * 3 + 2] = clamp8(y + 1.772f * cb);
        }
        return img;
    }
};

} // namespace jpg

inline JpegImage readJpeg(const std::string& path)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f)
        jpg::bad(path, "cannot open");
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0)
        bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    jpg::Decoder d(path, bytes);
    return d.run();
}

} // namespace igh
