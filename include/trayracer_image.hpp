// trayracer_image.hpp -- TIFF, baseline-JPEG and OpenEXR decoding for the C++ scene loader (trayracer_loader.hpp), header-only C++17.
//
// The reference opens every texture and sky image with `image::open(path)?.into_rgb8()` (src/scene/object.rs:31-38,
// src/scene/skybox.rs:56-63): any format the `image` crate reads, reduced to 8-bit RGB.  Its own test scene uses TIFF
// (uncompressed and LZW with horizontal differencing) and baseline 4:2:0 JPEG textures (res/test/textures/).  These decoders
// cover what those files need and the common neighbours:
//   TIFF  -- little / big endian, strips, 8 bits per sample, chunky; compression none, LZW (predictor 1 / 2), PackBits;
//            photometric BlackIsZero / WhiteIsZero / RGB / palette; extra samples (alpha) dropped as into_rgb8() drops them.
//            Lossless: the bytes equal any other correct decoder's (tests compare with Pillow).
//   JPEG  -- baseline sequential Huffman (SOF0 / SOF1 8-bit), 1 or 3 components, any sampling factors up to 2x2 per
//            component, restart intervals, JFIF YCbCr or Adobe RGB.  The arithmetic follows the IJG library's defaults
//            (integer "islow" inverse DCT, "fancy" triangle-filter chroma upsampling, 16-bit fixed-point colour conversion), which
//            is what Pillow (libjpeg-turbo) produces; the `image` crate's own decoder may differ by an LSB -- parity
//            unpinned against the crate, within 1 LSB of Pillow by test.
//   EXR   -- single-part scan-line / one-level tiled OpenEXR, HALF / FLOAT / UINT R G B, compression NONE, RLE, ZIPS, ZIP, PIZ,
//            PXR24 (the reference's skyboxes are EXR).  Parity unpinned: the reference's own EXR files are git-LFS stubs,
//            so the decoder is checked against files written by an independent encoder in tests/exr_writer.py only.
// Not decoded: tiled / 16-bit / float TIFF, progressive / arithmetic / 12-bit JPEG, B44 / DWA / deep / multi-part EXR --
// loading such a file fails like a failed image::open.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <zlib.h>

namespace trayracer {
namespace imgdec {

struct Rgb8 {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};
struct DecodeError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
// A header may claim any size (65535 x 65535 JPEG, 2^31 x 2^31 EXR): refuse what no texture or sky of this renderer is --
// above 2^28 pixels (16384 x 16384; the reference's largest sky is 8192 x 4096) -- before anything is allocated for it.
constexpr uint64_t MAX_PIXELS = 1ull << 28;
inline void check_dimensions(uint64_t w, uint64_t h) {
    if (w == 0 || h == 0 || w > (1ull << 20) || h > (1ull << 20) || w * h > MAX_PIXELS)
        throw DecodeError("image dimensions " + std::to_string(w) + " x " + std::to_string(h) + " out of range");
}

// ---------------------------------------------------------------------------------------------------------------
// TIFF
// ---------------------------------------------------------------------------------------------------------------
namespace tiff {
struct Reader {
    const uint8_t* p;
    size_t n;
    bool le;
    uint16_t u16(size_t o) const {
        if (o + 2 > n) throw DecodeError("truncated TIFF");
        return le ? (uint16_t)(p[o] | (p[o + 1] << 8)) : (uint16_t)((p[o] << 8) | p[o + 1]);
    }
    uint32_t u32(size_t o) const {
        if (o + 4 > n) throw DecodeError("truncated TIFF");
        return le ? ((uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24))
                  : (((uint32_t)p[o] << 24) | ((uint32_t)p[o + 1] << 16) | ((uint32_t)p[o + 2] << 8) | (uint32_t)p[o + 3]);
    }
};
struct Entry {
    uint16_t tag = 0, type = 0;
    uint32_t count = 0;
    size_t value_off = 0;   // where the values live (inside the entry when they fit 4 bytes)
};
inline std::vector<uint32_t> values(const Reader& r, const Entry& e) {
    std::vector<uint32_t> v;
    const size_t sz = e.type == 3 ? 2 : e.type == 4 ? 4 : e.type == 1 ? 1 : 0;
    if (sz == 0) throw DecodeError("unsupported TIFF field type");
    if (e.count > (1u << 24)) throw DecodeError("TIFF field too long");
    for (uint32_t k = 0; k < e.count; ++k) {
        size_t o = e.value_off + k * sz;
        if (o + sz > r.n) throw DecodeError("truncated TIFF");
        v.push_back(sz == 2 ? r.u16(o) : sz == 4 ? r.u32(o) : r.p[o]);
    }
    return v;
}

// TIFF 6.0 section 13: MSB-first codes from 9 bits, ClearCode 256, EndOfInformation 257, the width grows one code EARLY
inline void lzw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t want) {
    struct Item { int32_t prev; uint8_t ch, first; uint32_t len; };
    std::vector<Item> tab(4096);
    for (int i = 0; i < 256; ++i) tab[i] = Item{-1, (uint8_t)i, (uint8_t)i, 1};
    uint32_t next = 258, width = 9;
    uint64_t acc = 0;
    int bits = 0;
    size_t pos = 0;
    int32_t prev = -1;
    std::vector<uint8_t> tmp;
    auto emit = [&](int32_t code) {
        const uint32_t len = tab[code].len;
        size_t base = out.size();
        out.resize(base + len);
        for (int32_t c = code; c >= 0; c = tab[c].prev) out[base + tab[c].len - 1] = tab[c].ch;
    };
    while (out.size() < want) {
        while (bits < (int)width) {
            if (pos >= n) return;   // ran out of data: the caller checks the size
            acc = (acc << 8) | src[pos++];
            bits += 8;
        }
        const uint32_t code = (uint32_t)((acc >> (bits - width)) & ((1u << width) - 1));
        bits -= (int)width;
        if (code == 257) break;
        if (code == 256) {
            next = 258; width = 9; prev = -1;
            continue;
        }
        if (prev < 0) {
            if (code >= 256) throw DecodeError("corrupt TIFF LZW stream");
            emit((int32_t)code);
        } else if (code < next) {
            emit((int32_t)code);
            if (next < 4096) { tab[next] = Item{prev, tab[code].first, tab[prev].first, tab[prev].len + 1}; ++next; }
        } else if (code == next && next < 4096) {
            tab[next] = Item{prev, tab[prev].first, tab[prev].first, tab[prev].len + 1};
            ++next;
            emit((int32_t)code);
        } else {
            throw DecodeError("corrupt TIFF LZW stream");
        }
        prev = (int32_t)code;
        if (next + 1 >= (1u << width) && width < 12) ++width;   // "early change"
    }
}

inline void packbits(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t want) {
    size_t pos = 0;
    while (pos < n && out.size() < want) {
        const int8_t c = (int8_t)src[pos++];
        if (c >= 0) {
            size_t k = (size_t)c + 1;
            if (pos + k > n) throw DecodeError("corrupt TIFF PackBits stream");
            out.insert(out.end(), src + pos, src + pos + k);
            pos += k;
        } else if (c != -128) {
            if (pos >= n) throw DecodeError("corrupt TIFF PackBits stream");
            out.insert(out.end(), (size_t)(1 - c), src[pos++]);
        }
    }
}

inline Rgb8 decode(const std::string& d) {
    Reader r{reinterpret_cast<const uint8_t*>(d.data()), d.size(), true};
    if (d.size() < 8) throw DecodeError("not a TIFF file");
    if (d[0] == 'I' && d[1] == 'I') r.le = true;
    else if (d[0] == 'M' && d[1] == 'M') r.le = false;
    else throw DecodeError("not a TIFF file");
    if (r.u16(2) != 42) throw DecodeError("not a TIFF file (BigTIFF is not supported)");
    const size_t ifd = r.u32(4);
    const uint16_t n_entries = r.u16(ifd);
    uint32_t width = 0, height = 0, compression = 1, photometric = 2, spp = 1, rows_per_strip = 0xFFFFFFFFu, planar = 1, predictor = 1;
    std::vector<uint32_t> bps{1}, offsets, counts, colormap;
    bool tiled = false;
    for (uint16_t k = 0; k < n_entries; ++k) {
        const size_t o = ifd + 2 + (size_t)k * 12;
        Entry e;
        e.tag = r.u16(o); e.type = r.u16(o + 2); e.count = r.u32(o + 4);
        const size_t sz = e.type == 3 ? 2 : e.type == 4 ? 4 : e.type == 1 ? 1 : 0;
        e.value_off = (sz != 0 && (size_t)e.count * sz <= 4) ? o + 8 : r.u32(o + 8);
        auto first = [&]() { auto v = values(r, e); if (v.empty()) throw DecodeError("empty TIFF field"); return v[0]; };
        switch (e.tag) {
            case 256: width = first(); break;
            case 257: height = first(); break;
            case 258: bps = values(r, e); break;
            case 259: compression = first(); break;
            case 262: photometric = first(); break;
            case 273: offsets = values(r, e); break;
            case 277: spp = first(); break;
            case 278: rows_per_strip = first(); break;
            case 279: counts = values(r, e); break;
            case 284: planar = first(); break;
            case 317: predictor = first(); break;
            case 320: colormap = values(r, e); break;
            case 322: case 323: case 324: case 325: tiled = true; break;
            default: break;
        }
    }
    if (tiled) throw DecodeError("unsupported TIFF (tiles)");
    if (width == 0 || height == 0 || width > 65535 || height > 65535) throw DecodeError("unsupported TIFF size");
    for (uint32_t b : bps)
        if (b != 8) throw DecodeError("unsupported TIFF (only 8 bits per sample)");
    if (planar != 1 && spp > 1) throw DecodeError("unsupported TIFF (planar configuration 2)");
    if (compression != 1 && compression != 5 && compression != 32773) throw DecodeError("unsupported TIFF compression " + std::to_string(compression));
    if (predictor != 1 && predictor != 2) throw DecodeError("unsupported TIFF predictor");
    if (spp < 1 || spp > 8) throw DecodeError("unsupported TIFF samples per pixel");
    if (photometric == 2 && spp < 3) throw DecodeError("RGB TIFF with fewer than 3 samples");
    if (photometric > 3) throw DecodeError("unsupported TIFF photometric interpretation " + std::to_string(photometric));
    if (photometric == 3 && colormap.size() < 768) throw DecodeError("palette TIFF without a ColorMap");
    if (offsets.empty() || offsets.size() != counts.size()) throw DecodeError("TIFF without strips");
    if (rows_per_strip == 0) throw DecodeError("bad TIFF RowsPerStrip");
    rows_per_strip = std::min(rows_per_strip, height);
    const size_t row_bytes = (size_t)width * spp;

    check_dimensions(width, height);
    Rgb8 im;
    im.width = width; im.height = height;
    im.rgb.resize((size_t)width * height * 3);
    std::vector<uint8_t> buf;
    uint32_t y0 = 0;
    for (size_t s = 0; s < offsets.size() && y0 < height; ++s) {
        const uint32_t rows = std::min(rows_per_strip, height - y0);
        const size_t want = row_bytes * rows;
        if ((size_t)offsets[s] + counts[s] > d.size()) throw DecodeError("truncated TIFF strip");
        const uint8_t* src = r.p + offsets[s];
        buf.clear();
        if (compression == 1) buf.assign(src, src + std::min<size_t>(counts[s], want));
        else if (compression == 5) lzw(src, counts[s], buf, want);
        else packbits(src, counts[s], buf, want);
        if (buf.size() < want) throw DecodeError("short TIFF strip");
        for (uint32_t y = 0; y < rows; ++y) {
            uint8_t* row = buf.data() + (size_t)y * row_bytes;
            if (predictor == 2)   // horizontal differencing: each sample adds the same sample of the pixel to its left
                for (size_t x = spp; x < row_bytes; ++x) row[x] = (uint8_t)(row[x] + row[x - spp]);
            uint8_t* dst = im.rgb.data() + (size_t)(y0 + y) * width * 3;
            for (uint32_t x = 0; x < width; ++x) {
                const uint8_t* px = row + (size_t)x * spp;
                uint8_t R, G, B;
                if (photometric == 2) { R = px[0]; G = px[1]; B = px[2]; }
                else if (photometric == 3) {   // 16-bit colour map entries: the high byte, as libtiff-based readers do
                    R = (uint8_t)(colormap[px[0]] >> 8); G = (uint8_t)(colormap[256 + px[0]] >> 8); B = (uint8_t)(colormap[512 + px[0]] >> 8);
                } else { R = G = B = photometric == 0 ? (uint8_t)(255 - px[0]) : px[0]; }
                dst[x * 3] = R; dst[x * 3 + 1] = G; dst[x * 3 + 2] = B;
            }
        }
        y0 += rows;
    }
    if (y0 < height) throw DecodeError("TIFF strips do not cover the image");
    return im;
}
}   // namespace tiff

// ---------------------------------------------------------------------------------------------------------------
// baseline JPEG (ITU T.81), with the IJG library's default arithmetic
// ---------------------------------------------------------------------------------------------------------------
namespace jpeg {
static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int32_t mincode[17], maxcode[18], valptr[17];
    void build() {
        int32_t code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;       // blocks allocated (padded to whole MCUs)
    int width = 0, height = 0;            // real downsampled size: ceil(image * h / hmax)
    std::vector<uint8_t> plane;           // blocks_w * 8 by blocks_h * 8 samples
    int pred = 0;
};

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t acc = 0;
    int bits = 0;
    bool hit_marker = false;
    void fill() {
        while (bits <= 24) {
            uint8_t b = 0;
            if (!hit_marker && pos < n) {
                b = p[pos];
                if (b == 0xFF) {
                    if (pos + 1 < n && p[pos + 1] == 0x00) pos += 2;
                    else { hit_marker = true; b = 0; }   // a marker: feed zeros, as the IJG decoder does
                } else ++pos;
            }
            acc |= (uint32_t)b << (24 - bits);
            bits += 8;
        }
    }
    int get(int k) {
        if (k == 0) return 0;
        if (bits < k) fill();
        int v = (int)(acc >> (32 - k));
        acc <<= k;
        bits -= k;
        return v;
    }
    void reset() { acc = 0; bits = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const Huff& h) {
    int32_t code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.get(1);
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + (code - h.mincode[l])];
    }
    throw DecodeError("corrupt JPEG Huffman code");
}
inline int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

// jidctint.c (JDCT_ISLOW): Loeffler-Ligtenberg-Moschytz, 13-bit constants, two extra bits kept between the passes
inline void idct_islow(const int32_t* in /* dequantised, natural order */, uint8_t* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int32_t F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                      F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
                      F_2_562915447 = 20995, F_3_072711026 = 25172;
    auto descale = [](int64_t x, int n) { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); };
    int32_t ws[64];
    auto pass = [&](int32_t i0, int32_t i1, int32_t i2, int32_t i3, int32_t i4, int32_t i5, int32_t i6, int32_t i7, int shift, int32_t* o) {
        int64_t z2 = i2, z3 = i6;
        int64_t z1 = (z2 + z3) * F_0_541196100;
        int64_t tmp2 = z1 + z3 * (-F_1_847759065);
        int64_t tmp3 = z1 + z2 * F_0_765366865;
        z2 = i0; z3 = i4;
        int64_t tmp0 = (z2 + z3) * ((int64_t)1 << CB);
        int64_t tmp1 = (z2 - z3) * ((int64_t)1 << CB);
        int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = i7; tmp1 = i5; tmp2 = i3; tmp3 = i1;
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        int64_t z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        o[0] = descale(tmp10 + tmp3, shift); o[7] = descale(tmp10 - tmp3, shift);
        o[1] = descale(tmp11 + tmp2, shift); o[6] = descale(tmp11 - tmp2, shift);
        o[2] = descale(tmp12 + tmp1, shift); o[5] = descale(tmp12 - tmp1, shift);
        o[3] = descale(tmp13 + tmp0, shift); o[4] = descale(tmp13 - tmp0, shift);
    };
    for (int c = 0; c < 8; ++c) {   // pass 1: columns
        int32_t o[8];
        pass(in[c], in[8 + c], in[16 + c], in[24 + c], in[32 + c], in[40 + c], in[48 + c], in[56 + c], CB - P1, o);
        for (int k = 0; k < 8; ++k) ws[k * 8 + c] = o[k];
    }
    for (int rr = 0; rr < 8; ++rr) {   // pass 2: rows, remove the extra scaling and the DCT's factor of 8, level shift
        int32_t o[8];
        const int32_t* w = ws + rr * 8;
        pass(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], CB + P1 + 3, o);
        for (int k = 0; k < 8; ++k) {
            int32_t v = o[k] + 128;
            out[rr * stride + k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

inline Rgb8 decode(const std::string& d) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(d.data());
    const size_t n = d.size();
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) throw DecodeError("not a JPEG file");
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int width = 0, height = 0, restart_interval = 0;
    int adobe_transform = -1;
    bool have_frame = false;
    size_t pos = 2;
    auto be16 = [&](size_t o) -> int { if (o + 2 > n) throw DecodeError("truncated JPEG"); return (p[o] << 8) | p[o + 1]; };

    for (;;) {
        while (pos < n && p[pos] != 0xFF) ++pos;
        while (pos < n && p[pos] == 0xFF) ++pos;
        if (pos >= n) throw DecodeError("JPEG without image data");
        const uint8_t m = p[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) throw DecodeError("JPEG without image data");
        const int len = be16(pos);
        if (len < 2 || pos + (size_t)len > n) throw DecodeError("truncated JPEG segment");
        const uint8_t* s = p + pos + 2;
        const int sl = len - 2;
        if (m == 0xDB) {
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                ++o;
                if (tq > 3 || o + (pq ? 128 : 64) > sl) throw DecodeError("bad JPEG quantisation table");
                for (int k = 0; k < 64; ++k) {
                    qt[tq][ZIGZAG[k]] = pq ? (uint16_t)((s[o] << 8) | s[o + 1]) : s[o];
                    o += pq ? 2 : 1;
                }
                qt_present[tq] = true;
            }
        } else if (m == 0xC4) {
            int o = 0;
            while (o < sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                ++o;
                if (tc > 1 || th > 3 || o + 16 > sl) throw DecodeError("bad JPEG Huffman table");
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = s[o + l - 1]; total += h.bits[l]; }
                o += 16;
                if (total > 256 || o + total > sl) throw DecodeError("bad JPEG Huffman table");
                std::memcpy(h.vals, s + o, (size_t)total);
                o += total;
                h.present = true;
                h.build();
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) throw DecodeError("unsupported JPEG sample precision");
            height = (s[1] << 8) | s[2];
            width = (s[3] << 8) | s[4];
            const int nc = s[5];
            if (width <= 0 || height <= 0 || (nc != 1 && nc != 3) || sl < 6 + 3 * nc) throw DecodeError("unsupported JPEG frame");
            check_dimensions((uint64_t)width, (uint64_t)height);
            comps.assign((size_t)nc, Component());
            for (int c = 0; c < nc; ++c) {
                comps[c].id = s[6 + 3 * c];
                comps[c].h = s[7 + 3 * c] >> 4;
                comps[c].v = s[7 + 3 * c] & 15;
                comps[c].tq = s[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 2 || comps[c].v < 1 || comps[c].v > 2 || comps[c].tq > 3) throw DecodeError("unsupported JPEG sampling factors");
            }
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            throw DecodeError("unsupported JPEG process (only baseline sequential Huffman)");
        } else if (m == 0xDD) {
            if (sl >= 2) restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xEE) {
            if (sl >= 12 && std::memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
        } else if (m == 0xDA) {
            if (!have_frame) throw DecodeError("JPEG scan before the frame header");
            const int ns = s[0];
            if (ns != (int)comps.size() || sl < 1 + 2 * ns + 3) throw DecodeError("unsupported JPEG (a scan per component)");
            for (int k = 0; k < ns; ++k) {
                const int cid = s[1 + 2 * k];
                bool found = false;
                for (auto& c : comps)
                    if (c.id == cid) { c.td = s[2 + 2 * k] >> 4; c.ta = s[2 + 2 * k] & 15; found = true; }
                if (!found) throw DecodeError("JPEG scan names an unknown component");
                // Td / Ta index dc[4] / ac[4] below: a baseline scan may only name tables 0..3 (T.81 B.2.3)
                if ((s[2 + 2 * k] >> 4) > 3 || (s[2 + 2 * k] & 15) > 3) throw DecodeError("bad JPEG table selector");
            }
            pos += (size_t)len;
            break;
        }
        pos += (size_t)len;
    }

    int hmax = 1, vmax = 1;
    for (auto& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
    if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }   // a single-component scan is never interleaved
    const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
    for (auto& c : comps) {
        if (!qt_present[c.tq] || !dc[c.td].present || !ac[c.ta].present) throw DecodeError("JPEG component without its tables");
        c.blocks_w = mcus_x * c.h;
        c.blocks_h = mcus_y * c.v;
        c.width = (width * c.h + hmax - 1) / hmax;
        c.height = (height * c.v + vmax - 1) / vmax;
        c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
    }

    // ---- entropy-coded data: one interleaved scan ----
    BitReader br{p, n};
    br.pos = pos;
    int32_t coef[64];
    int to_restart = restart_interval;
    for (int my = 0; my < mcus_y; ++my)
        for (int mx = 0; mx < mcus_x; ++mx) {
            if (restart_interval && to_restart == 0) {
                // byte-align, expect RSTn
                br.reset();
                while (br.pos + 1 < n && !(p[br.pos] == 0xFF && p[br.pos + 1] >= 0xD0 && p[br.pos + 1] <= 0xD7)) ++br.pos;
                if (br.pos + 1 < n) br.pos += 2;
                for (auto& c : comps) c.pred = 0;
                to_restart = restart_interval;
            }
            for (auto& c : comps)
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx) {
                        std::memset(coef, 0, sizeof coef);
                        const int t = decode_symbol(br, dc[c.td]);
                        if (t > 11) throw DecodeError("corrupt JPEG DC coefficient");
                        c.pred += extend(br.get(t), t);
                        coef[0] = c.pred * (int32_t)qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            const int rs = decode_symbol(br, ac[c.ta]);
                            const int rr = rs >> 4, ss = rs & 15;
                            if (ss == 0) {
                                if (rr == 15) { k += 16; continue; }
                                break;
                            }
                            k += rr;
                            if (k > 63) throw DecodeError("corrupt JPEG AC coefficient");
                            coef[ZIGZAG[k]] = extend(br.get(ss), ss) * (int32_t)qt[c.tq][ZIGZAG[k]];
                            ++k;
                        }
                        const int stride = c.blocks_w * 8;
                        idct_islow(coef, c.plane.data() + (size_t)(my * c.v + by) * 8 * stride + (size_t)(mx * c.h + bx) * 8, stride);
                    }
            if (restart_interval) --to_restart;
        }

    // ---- upsampling to full resolution (jdsample.c: "fancy" triangle filters; edge samples replicate) ----
    auto upsample = [&](const Component& c) {
        std::vector<uint8_t> full((size_t)width * height);
        const int stride = c.blocks_w * 8;
        auto src = [&](int y) { return c.plane.data() + (size_t)std::min(std::max(y, 0), c.height - 1) * stride; };
        const int cw = c.width;
        const bool h2 = c.h * 2 == hmax, v2 = c.v * 2 == vmax;
        if ((c.h != hmax && !h2) || (c.v != vmax && !v2)) throw DecodeError("unsupported JPEG sampling ratio");
        std::vector<int32_t> colsum((size_t)cw);
        std::vector<uint8_t> wide((size_t)cw * 2 + 2);
        for (int y = 0; y < height; ++y) {
            uint8_t* dst = full.data() + (size_t)y * width;
            if (!h2 && !v2) { std::memcpy(dst, src(y), (size_t)width); continue; }
            if (h2 && !v2) {   // h2v1_fancy_upsample
                const uint8_t* in = src(y);
                if (cw == 1) { wide[0] = wide[1] = in[0]; }
                else {
                    wide[0] = in[0];
                    wide[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                    for (int x = 1; x < cw - 1; ++x) {
                        wide[2 * x] = (uint8_t)((in[x] * 3 + in[x - 1] + 1) >> 2);
                        wide[2 * x + 1] = (uint8_t)((in[x] * 3 + in[x + 1] + 2) >> 2);
                    }
                    wide[2 * (cw - 1)] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
                    wide[2 * (cw - 1) + 1] = in[cw - 1];
                }
                std::memcpy(dst, wide.data(), (size_t)width);
                continue;
            }
            // vertical 2x: the nearer input row counts 3/4, the farther 1/4
            const int iy = y >> 1;
            const uint8_t* in0 = src(iy);
            const uint8_t* in1 = src((y & 1) ? iy + 1 : iy - 1);
            if (!h2) {   // h1v2_fancy_upsample
                for (int x = 0; x < width; ++x) dst[x] = (uint8_t)((in0[x] * 3 + in1[x] + ((y & 1) ? 2 : 1)) >> 2);
                continue;
            }
            // h2v2_fancy_upsample
            for (int x = 0; x < cw; ++x) colsum[x] = in0[x] * 3 + in1[x];
            if (cw == 1) { wide[0] = (uint8_t)((colsum[0] * 4 + 8) >> 4); wide[1] = (uint8_t)((colsum[0] * 4 + 7) >> 4); }
            else {
                wide[0] = (uint8_t)((colsum[0] * 4 + 8) >> 4);
                wide[1] = (uint8_t)((colsum[0] * 3 + colsum[1] + 7) >> 4);
                for (int x = 1; x < cw - 1; ++x) {
                    wide[2 * x] = (uint8_t)((colsum[x] * 3 + colsum[x - 1] + 8) >> 4);
                    wide[2 * x + 1] = (uint8_t)((colsum[x] * 3 + colsum[x + 1] + 7) >> 4);
                }
                wide[2 * (cw - 1)] = (uint8_t)((colsum[cw - 1] * 3 + colsum[cw - 2] + 8) >> 4);
                wide[2 * (cw - 1) + 1] = (uint8_t)((colsum[cw - 1] * 4 + 7) >> 4);
            }
            std::memcpy(dst, wide.data(), (size_t)width);
        }
        return full;
    };

    Rgb8 im;
    im.width = (uint32_t)width; im.height = (uint32_t)height;
    im.rgb.resize((size_t)width * height * 3);
    if (comps.size() == 1) {
        const std::vector<uint8_t> y = upsample(comps[0]);
        for (size_t i = 0; i < y.size(); ++i) im.rgb[i * 3] = im.rgb[i * 3 + 1] = im.rgb[i * 3 + 2] = y[i];
        return im;
    }
    const std::vector<uint8_t> c0 = upsample(comps[0]), c1 = upsample(comps[1]), c2 = upsample(comps[2]);
    const bool ycc = adobe_transform != 0;   // JFIF (and Adobe transform 1) store YCbCr; Adobe transform 0 stores RGB
    if (!ycc) {
        for (size_t i = 0; i < c0.size(); ++i) { im.rgb[i * 3] = c0[i]; im.rgb[i * 3 + 1] = c1[i]; im.rgb[i * 3 + 2] = c2[i]; }
        return im;
    }
    // jdcolor.c: 16-bit fixed point, tables built once
    int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
        const int32_t x = i - 128;
        cr_r[i] = (91881 * x + 32768) >> 16;     // 1.40200
        cb_b[i] = (116130 * x + 32768) >> 16;    // 1.77200
        cr_g[i] = -46802 * x;                    // 0.71414
        cb_g[i] = -22554 * x + 32768;            // 0.34414 (+ the rounding half)
    }
    auto clamp8 = [](int32_t v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    for (size_t i = 0; i < c0.size(); ++i) {
        const int32_t y = c0[i], cb = c1[i], cr = c2[i];
        im.rgb[i * 3] = clamp8(y + cr_r[cr]);
        im.rgb[i * 3 + 1] = clamp8(y + ((cb_g[cb] + cr_g[cr]) >> 16));
        im.rgb[i * 3 + 2] = clamp8(y + cb_b[cb]);
    }
    return im;
}
}   // namespace jpeg

// ---------------------------------------------------------------------------------------------------------------
// OpenEXR (the reference's skyboxes, res/skyboxes/*.exr; skybox.rs:56-63: image::open(path)?.into_rgb8()).
// Single-part scan-line and one-level tiled files, HALF / FLOAT / UINT channels R, G, B (others ignored), compression NONE,
// RLE, ZIPS, ZIP, PIZ, PXR24; restated from the published file-format documents ("OpenEXR File Layout", "Technical
// Introduction") and the open-source library's PIZ description.  Reduced to 8 bits like the image crate's
// Rgb32F -> Rgb8 conversion: round(clamp(v, 0, 1) * 255), linear, no tone mapping.
// ---------------------------------------------------------------------------------------------------------------
namespace exr {
struct Cursor {
    const uint8_t* p;
    size_t n, o = 0;
    void need(size_t k) const { if (o + k > n) throw DecodeError("truncated EXR"); }
    uint8_t u8() { need(1); return p[o++]; }
    uint32_t u32() { need(4); uint32_t v; std::memcpy(&v, p + o, 4); o += 4; return v; }
    int32_t i32() { return (int32_t)u32(); }
    uint64_t u64() { need(8); uint64_t v; std::memcpy(&v, p + o, 8); o += 8; return v; }
    std::string cstr() {
        std::string s;
        for (;;) { need(1); char c = (char)p[o++]; if (!c) break; s.push_back(c); if (s.size() > 255) throw DecodeError("corrupt EXR header"); }
        return s;
    }
};
struct Channel {
    std::string name;
    int32_t type = 1;   // 0 UINT, 1 HALF, 2 FLOAT
    int32_t xs = 1, ys = 1;
    size_t bytes() const { return type == 1 ? 2 : 4; }
};

inline float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = s;
        else {   // subnormal: renormalise
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 1024u)) { mm <<= 1; ++sh; }
            bits = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) bits = s | 0x7F800000u | (m << 13);
    else bits = s | ((e + 112u) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// ---- RLE / ZIP: byte streams with a delta predictor and an even/odd split --------------------------------------------
inline void unpredict_and_interleave(std::vector<uint8_t>& t, std::vector<uint8_t>& out) {
    for (size_t i = 1; i < t.size(); ++i) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
    out.resize(t.size());
    const size_t half = (t.size() + 1) / 2;
    for (size_t i = 0; i < t.size(); ++i) out[i] = (i & 1) ? t[half + i / 2] : t[i / 2];
}
inline void rle_decode(const uint8_t* src, size_t n, size_t want, std::vector<uint8_t>& out) {
    out.clear();
    size_t i = 0;
    while (i < n) {
        const int c = (int8_t)src[i++];
        if (c < 0) {
            const size_t k = (size_t)(-c);
            if (i + k > n || out.size() + k > want) throw DecodeError("corrupt EXR RLE data");
            out.insert(out.end(), src + i, src + i + k);
            i += k;
        } else {
            if (i >= n || out.size() + (size_t)c + 1 > want) throw DecodeError("corrupt EXR RLE data");
            out.insert(out.end(), (size_t)c + 1, src[i++]);
        }
    }
    if (out.size() != want) throw DecodeError("short EXR RLE data");
}
inline void zlib_decode(const uint8_t* src, size_t n, size_t want, std::vector<uint8_t>& out) {
    out.resize(want);
    uLongf got = (uLongf)want;
    if (::uncompress(out.data(), &got, src, (uLong)n) != Z_OK || got != want) throw DecodeError("corrupt EXR zlib data");
}

// ---- PIZ: value bitmap -> look-up table, Huffman-coded 16-bit symbols, 2-D Haar-like wavelet -----------------------
namespace piz {
constexpr int HUF_ENCBITS = 16, HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1;
constexpr int SHORT_ZEROCODE_RUN = 59, LONG_ZEROCODE_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN;
struct Bits {
    const uint8_t* p;
    size_t n, o = 0;
    uint64_t acc = 0;
    int cnt = 0;
    uint32_t get(int k) {   // most significant bit first
        while (cnt < k) {
            if (o >= n) throw DecodeError("truncated EXR PIZ data");
            acc = (acc << 8) | p[o++];
            cnt += 8;
        }
        cnt -= k;
        return (uint32_t)((acc >> cnt) & ((1ull << k) - 1ull));
    }
};
// code lengths of the symbols im..iM, 6 bits each, zero runs packed
inline void unpack_lengths(const uint8_t* p, size_t n, uint32_t im, uint32_t iM, std::vector<uint8_t>& len, size_t* used) {
    Bits b{p, n};
    for (uint32_t i = im; i <= iM; ++i) {
        const uint32_t l = b.get(6);
        if (l == (uint32_t)LONG_ZEROCODE_RUN) {
            uint32_t run = b.get(8) + SHORTEST_LONG_RUN;
            if (i + run > iM + 1) throw DecodeError("corrupt EXR PIZ code table");
            while (run--) len[i++] = 0;
            --i;
        } else if (l >= (uint32_t)SHORT_ZEROCODE_RUN) {
            uint32_t run = l - SHORT_ZEROCODE_RUN + 2;
            if (i + run > iM + 1) throw DecodeError("corrupt EXR PIZ code table");
            while (run--) len[i++] = 0;
            --i;
        } else len[i] = (uint8_t)l;
    }
    *used = b.o;
}
inline void huf_uncompress(const uint8_t* src, size_t n, uint16_t* out, size_t n_out) {
    if (n < 20) { if (n_out) throw DecodeError("truncated EXR PIZ data"); return; }
    uint32_t im, iM, table_len, n_bits;
    std::memcpy(&im, src, 4); std::memcpy(&iM, src + 4, 4); std::memcpy(&table_len, src + 8, 4); std::memcpy(&n_bits, src + 12, 4);
    if ((size_t)table_len > n - 20) throw DecodeError("corrupt EXR PIZ header");
    if (im >= (uint32_t)HUF_ENCSIZE || iM >= (uint32_t)HUF_ENCSIZE || im > iM) throw DecodeError("corrupt EXR PIZ header");
    std::vector<uint8_t> len(HUF_ENCSIZE, 0);
    size_t used = 0;
    unpack_lengths(src + 20, table_len, im, iM, len, &used);
    used = table_len;
    // canonical codes: lengths 58 .. 1, the longest codes numerically first (start[l]); inside a length by symbol
    uint64_t count[59] = {0}, start[59] = {0};
    for (uint32_t i = im; i <= iM; ++i) count[len[i]]++;
    uint64_t c = 0;
    for (int l = 58; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; start[l] = c; c = nc; }
    // an over-subscribed table (Kraft sum above 1, e.g. three symbols of length 1) would give codes of length l that do not
    // fit l bits: refuse it here, before the codes index the look-up table below
    for (int l = 1; l <= 58; ++l)
        if (count[l] && start[l] + count[l] > (1ull << l)) throw DecodeError("corrupt EXR PIZ code table (over-subscribed)");
    std::vector<uint32_t> first(60, 0), syms;
    syms.reserve(iM - im + 1);
    for (int l = 1; l <= 58; ++l) {
        first[l] = (uint32_t)syms.size();
        if (count[l]) for (uint32_t i = im; i <= iM; ++i) if (len[i] == l) syms.push_back(i);
    }
    // a 12-bit table for the short codes, bit-by-bit beyond
    constexpr int FAST = 12;
    std::vector<uint32_t> fast(1u << FAST, 0);   // (symbol << 6) | length, 0 = longer code
    for (int l = 1; l <= FAST; ++l)
        for (uint64_t k = 0; k < count[l]; ++k) {
            const uint64_t code = start[l] + k;
            const uint32_t s = syms[first[l] + (uint32_t)k];
            const uint32_t base = (uint32_t)(code << (FAST - l));
            if ((size_t)base + (1u << (FAST - l)) > fast.size()) throw DecodeError("corrupt EXR PIZ code table");
            for (uint32_t f = 0; f < (1u << (FAST - l)); ++f) fast[base + f] = (s << 6) | (uint32_t)l;
        }
    const uint8_t* data = src + 20 + used;
    const size_t data_n = n - 20 - used;
    if ((size_t)(n_bits + 7) / 8 > data_n) throw DecodeError("truncated EXR PIZ data");
    const uint32_t rlc = iM;
    size_t bit = 0, o = 0;
    auto peek = [&](size_t at, int k) -> uint32_t {   // k <= 32 bits at bit offset `at`, zero-padded past the end
        uint64_t v = 0;
        const size_t byte = at >> 3;
        for (int j = 0; j < 6; ++j) v = (v << 8) | (byte + j < data_n ? data[byte + j] : 0);
        return (uint32_t)((v >> (48 - (at & 7) - k)) & ((1ull << k) - 1ull));
    };
    while (bit < n_bits) {
        uint32_t sym;
        const uint32_t f = fast[peek(bit, FAST)];
        if (f) { sym = f >> 6; bit += f & 63u; }
        else {
            uint64_t code = peek(bit, FAST);
            int l = FAST;
            bool found = false;
            while (l < 58) {
                ++l;
                code = (code << 1) | peek(bit + l - 1, 1);
                if (count[l] && code >= start[l] && code - start[l] < count[l]) { sym = syms[first[l] + (uint32_t)(code - start[l])]; found = true; break; }
            }
            if (!found) throw DecodeError("corrupt EXR PIZ code");
            bit += (size_t)l;
        }
        if (bit > n_bits) throw DecodeError("corrupt EXR PIZ data");
        if (sym == rlc) {
            if (bit + 8 > n_bits || o == 0) throw DecodeError("corrupt EXR PIZ run");
            const uint32_t run = peek(bit, 8);
            bit += 8;
            if (o + run > n_out) throw DecodeError("corrupt EXR PIZ run");
            const uint16_t v = out[o - 1];
            for (uint32_t k = 0; k < run; ++k) out[o++] = v;
        } else {
            if (o >= n_out) throw DecodeError("corrupt EXR PIZ data (too many symbols)");
            out[o++] = (uint16_t)sym;
        }
    }
    if (o != n_out) throw DecodeError("short EXR PIZ data");
}
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int16_t ls = (int16_t)l, hs = (int16_t)h;
    const int hi = hs, ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hi);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xFFFF, aa = (d + bb - 0x8000) & 0xFFFF;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}
inline void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1;
    p2 = p;
    p >>= 1;
    auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) wdec14(l, h, a, b); else wdec16(l, h, a, b); };
    while (p >= 1) {
        uint16_t* py = in;
        uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py;
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
            if (nx & p) {
                uint16_t* p10 = px + oy1;
                dec(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t* px = py;
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
// one block: `ny` lines of `nx` pixels, channels in file order; the result has the uncompressed block's layout
inline void decode_block(const uint8_t* src, size_t n, const std::vector<Channel>& ch, int nx, int ny, std::vector<uint8_t>& out) {
    size_t words = 0;
    for (const Channel& c : ch) words += (size_t)nx * ny * (c.bytes() / 2);
    std::vector<uint16_t> tmp(words);
    if (n < 4) throw DecodeError("truncated EXR PIZ block");
    uint16_t min_nz, max_nz;
    std::memcpy(&min_nz, src, 2); std::memcpy(&max_nz, src + 2, 2);
    size_t o = 4;
    std::vector<uint8_t> bitmap(8192, 0);
    if (max_nz >= 8192) throw DecodeError("corrupt EXR PIZ bitmap");
    if (min_nz <= max_nz) {
        const size_t k = (size_t)max_nz - min_nz + 1;
        if (o + k > n) throw DecodeError("truncated EXR PIZ bitmap");
        std::memcpy(bitmap.data() + min_nz, src + o, k);
        o += k;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (o + 4 > n) throw DecodeError("truncated EXR PIZ block");
    int32_t length;
    std::memcpy(&length, src + o, 4);
    o += 4;
    if (length < 0 || o + (size_t)length > n) throw DecodeError("corrupt EXR PIZ block length");
    huf_uncompress(src + o, (size_t)length, tmp.data(), words);
    std::vector<uint16_t*> start(ch.size());
    uint16_t* q = tmp.data();
    for (size_t c = 0; c < ch.size(); ++c) {
        start[c] = q;
        const int size = (int)(ch[c].bytes() / 2);
        for (int j = 0; j < size; ++j) wav2_decode(q + j, nx, size, ny, nx * size, max_value);
        q += (size_t)nx * ny * size;
    }
    for (uint16_t& v : tmp) v = lut[v];
    out.resize(words * 2);
    uint8_t* w = out.data();
    for (int y = 0; y < ny; ++y)
        for (size_t c = 0; c < ch.size(); ++c) {
            const size_t cnt = (size_t)nx * (ch[c].bytes() / 2);
            std::memcpy(w, start[c], cnt * 2);
            start[c] += cnt;
            w += cnt * 2;
        }
}
}   // namespace piz

// PXR24: zlib, then per line and channel byte planes (most significant first) of running sums; FLOAT keeps 24 bits
inline void pxr24_decode(const uint8_t* src, size_t n, const std::vector<Channel>& ch, int nx, int ny, std::vector<uint8_t>& out) {
    size_t packed = 0, full = 0;
    for (const Channel& c : ch) { packed += (size_t)nx * ny * (c.type == 2 ? 3 : c.bytes()); full += (size_t)nx * ny * c.bytes(); }
    std::vector<uint8_t> t(packed);
    uLongf got = (uLongf)packed;
    if (::uncompress(t.data(), &got, src, (uLong)n) != Z_OK || got != packed) throw DecodeError("corrupt EXR PXR24 data");
    out.resize(full);
    const uint8_t* p = t.data();
    uint8_t* w = out.data();
    for (int y = 0; y < ny; ++y)
        for (const Channel& c : ch) {
            if (c.type == 1) {
                const uint8_t *p0 = p, *p1 = p + nx;
                p += 2 * (size_t)nx;
                uint32_t px = 0;
                for (int x = 0; x < nx; ++x) { px += ((uint32_t)p0[x] << 8) | p1[x]; const uint16_t v = (uint16_t)px; std::memcpy(w, &v, 2); w += 2; }
            } else if (c.type == 2) {
                const uint8_t *p0 = p, *p1 = p + nx, *p2 = p + 2 * (size_t)nx;
                p += 3 * (size_t)nx;
                uint32_t px = 0;
                for (int x = 0; x < nx; ++x) { px += ((uint32_t)p0[x] << 24) | ((uint32_t)p1[x] << 16) | ((uint32_t)p2[x] << 8); std::memcpy(w, &px, 4); w += 4; }
            } else {
                const uint8_t *p0 = p, *p1 = p + nx, *p2 = p + 2 * (size_t)nx, *p3 = p + 3 * (size_t)nx;
                p += 4 * (size_t)nx;
                uint32_t px = 0;
                for (int x = 0; x < nx; ++x) { px += ((uint32_t)p0[x] << 24) | ((uint32_t)p1[x] << 16) | ((uint32_t)p2[x] << 8) | p3[x]; std::memcpy(w, &px, 4); w += 4; }
            }
        }
}

inline uint8_t to_u8(float v) {   // image crate: (v.clamp(0, 1) * 255).round() as u8
    if (!(v == v)) return 0;
    const float c = v < 0.0f ? 0.0f : v > 1.0f ? 1.0f : v;
    return (uint8_t)std::lround(c * 255.0f);
}

inline Rgb8 decode(const std::string& d) {
    Cursor r{reinterpret_cast<const uint8_t*>(d.data()), d.size()};
    if (r.u32() != 20000630u) throw DecodeError("not an OpenEXR file");
    const uint32_t version = r.u32();
    if ((version & 0xFFu) != 2u) throw DecodeError("unsupported OpenEXR version");
    if (version & 0x1800u) throw DecodeError("unsupported OpenEXR file (deep or multi-part)");
    const bool tiled = (version & 0x200u) != 0;
    std::vector<Channel> ch;
    int compression = -1, line_order = 0;
    int32_t dw[4] = {0, 0, -1, -1};
    uint32_t tile_w = 0, tile_h = 0, tile_mode = 0;
    bool have_dw = false;
    for (;;) {
        const std::string name = r.cstr();
        if (name.empty()) break;
        const std::string type = r.cstr();
        const int32_t size = r.i32();
        if (size < 0) throw DecodeError("corrupt EXR header");
        r.need((size_t)size);
        Cursor a{r.p + r.o, (size_t)size};
        r.o += (size_t)size;
        if (name == "channels" && type == "chlist") {
            for (;;) {
                Channel c;
                c.name = a.cstr();
                if (c.name.empty()) break;
                c.type = a.i32();
                a.u32();   // pLinear + reserved
                c.xs = a.i32();
                c.ys = a.i32();
                if (c.type < 0 || c.type > 2) throw DecodeError("corrupt EXR channel type");
                ch.push_back(c);
            }
        } else if (name == "compression") compression = a.u8();
        else if (name == "dataWindow" && type == "box2i") { for (int k = 0; k < 4; ++k) dw[k] = a.i32(); have_dw = true; }
        else if (name == "lineOrder") line_order = a.u8();
        else if (name == "tiles" && type == "tiledesc") { tile_w = a.u32(); tile_h = a.u32(); tile_mode = a.u8(); }
    }
    (void)line_order;   // blocks are found through the offset table and carry their own y
    if (ch.empty() || !have_dw || compression < 0) throw DecodeError("EXR header lacks channels, dataWindow or compression");
    const int64_t W = (int64_t)dw[2] - dw[0] + 1, H = (int64_t)dw[3] - dw[1] + 1;
    if (W <= 0 || H <= 0 || W > 65535 || H > 65535) throw DecodeError("unsupported EXR size");
    int idx[3] = {-1, -1, -1};
    for (size_t c = 0; c < ch.size(); ++c) {
        if (ch[c].xs != 1 || ch[c].ys != 1) throw DecodeError("unsupported EXR (sub-sampled channels)");
        if (ch[c].name == "R") idx[0] = (int)c;
        else if (ch[c].name == "G") idx[1] = (int)c;
        else if (ch[c].name == "B") idx[2] = (int)c;
    }
    if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0) throw DecodeError("EXR without R, G and B channels");
    int lines_per_block;
    switch (compression) {
        case 0: case 1: case 2: lines_per_block = 1; break;
        case 3: case 5: lines_per_block = 16; break;
        case 4: lines_per_block = 32; break;
        default: throw DecodeError("unsupported EXR compression " + std::to_string(compression) + " (B44 / DWA)");
    }
    size_t px_bytes = 0;   // bytes of one pixel over all channels
    std::vector<size_t> ch_off(ch.size());
    for (size_t c = 0; c < ch.size(); ++c) px_bytes += ch[c].bytes();
    check_dimensions((uint64_t)W, (uint64_t)H);
    Rgb8 im;
    im.width = (uint32_t)W; im.height = (uint32_t)H;
    im.rgb.assign((size_t)W * H * 3, 0);
    std::vector<uint8_t> raw, tmp;
    // decodes one block of nx x ny pixels whose top-left pixel is (x0, y0) in image coordinates
    auto block = [&](const uint8_t* src, size_t n, int x0, int y0, int nx, int ny) {
        const size_t want = px_bytes * (size_t)nx * ny;
        const uint8_t* data = src;
        if (n == want || compression == 0) {   // stored raw when compression did not help
            if (n < want) throw DecodeError("short EXR block");
        } else {
            if (compression == 1) { rle_decode(src, n, want, tmp); unpredict_and_interleave(tmp, raw); }
            else if (compression == 2 || compression == 3) { zlib_decode(src, n, want, tmp); unpredict_and_interleave(tmp, raw); }
            else if (compression == 4) piz::decode_block(src, n, ch, nx, ny, raw);
            else pxr24_decode(src, n, ch, nx, ny, raw);
            if (raw.size() != want) throw DecodeError("corrupt EXR block");
            data = raw.data();
        }
        for (int y = 0; y < ny; ++y) {
            const uint8_t* line = data + px_bytes * (size_t)nx * y;
            size_t off = 0;
            for (size_t c = 0; c < ch.size(); ++c) {
                const size_t bytes = ch[c].bytes();
                int comp = -1;
                for (int k = 0; k < 3; ++k) if (idx[k] == (int)c) comp = k;
                if (comp >= 0) {
                    uint8_t* o = im.rgb.data() + ((size_t)(y0 + y) * W + x0) * 3 + comp;
                    const uint8_t* s = line + off;
                    for (int x = 0; x < nx; ++x, o += 3, s += bytes) {
                        float v;
                        if (ch[c].type == 1) { uint16_t h; std::memcpy(&h, s, 2); v = half_to_float(h); }
                        else if (ch[c].type == 2) std::memcpy(&v, s, 4);
                        else { uint32_t u; std::memcpy(&u, s, 4); v = (float)u; }
                        *o = to_u8(v);
                    }
                }
                off += bytes * (size_t)nx;
            }
        }
    };
    if (!tiled) {
        const size_t n_blocks = (size_t)((H + lines_per_block - 1) / lines_per_block);
        std::vector<uint64_t> offs(n_blocks);
        for (uint64_t& o : offs) o = r.u64();
        for (uint64_t o : offs) {
            Cursor b{r.p, r.n, (size_t)o};
            if (o > r.n) throw DecodeError("corrupt EXR offset table");
            const int32_t y = b.i32();
            const int32_t size = b.i32();
            if (size < 0) throw DecodeError("corrupt EXR block");
            b.need((size_t)size);
            const int64_t y0 = (int64_t)y - dw[1];
            if (y0 < 0 || y0 >= H) throw DecodeError("corrupt EXR block position");
            const int ny = (int)std::min<int64_t>(lines_per_block, H - y0);
            block(b.p + b.o, (size_t)size, 0, (int)y0, (int)W, ny);
        }
    } else {
        if (tile_w == 0 || tile_h == 0 || tile_w > 65535 || tile_h > 65535) throw DecodeError("corrupt EXR tile description");
        (void)tile_mode;   // mip / rip maps: level (0, 0) comes first in the offset table and is all that is read
        const size_t tx = (size_t)((W + tile_w - 1) / tile_w), ty = (size_t)((H + tile_h - 1) / tile_h);
        std::vector<uint64_t> offs(tx * ty);
        for (uint64_t& o : offs) o = r.u64();
        for (uint64_t o : offs) {
            if (o > r.n) throw DecodeError("corrupt EXR offset table");
            Cursor b{r.p, r.n, (size_t)o};
            const int32_t txi = b.i32(), tyi = b.i32(), lx = b.i32(), ly = b.i32();
            const int32_t size = b.i32();
            if (size < 0 || lx != 0 || ly != 0 || txi < 0 || tyi < 0 || (size_t)txi >= tx || (size_t)tyi >= ty) throw DecodeError("corrupt EXR tile");
            b.need((size_t)size);
            const int x0 = (int)(txi * (int64_t)tile_w), y0 = (int)(tyi * (int64_t)tile_h);
            const int nx = (int)std::min<int64_t>(tile_w, W - x0), ny = (int)std::min<int64_t>(tile_h, H - y0);
            block(b.p + b.o, (size_t)size, x0, y0, nx, ny);
        }
    }
    return im;
}
}   // namespace exr

}   // namespace imgdec
}   // namespace trayracer
