// Minimal PNG reader/writer over zlib (libpng headers are not in this image). Upstream of the hot path; mirrors what the
// reference gets from the `png` 0.12 and `image` 0.19 crates (neither is vendored in the reference):
//   read_png_16bits   src/misc/helper.rs:13-36  — 16-bit grey, big-endian samples -> u16, row-major
//   read_luma8        src/bin/vors_track.rs:143 `image::open(..).to_luma()`:
//                       grey 8 passes through; grey 1/2/4 bits and palette images are expanded first (the png crate's EXPAND
//                       transformation: v * 255 / (2^bits - 1); palette -> RGB); grey+alpha drops alpha; RGB(A) -> luma by image
//                       0.19's `rgb_to_luma` (its src/color.rs): l = 0.2126 r + 0.7152 g + 0.0722 b in f32, NumCast to u8 =
//                       truncation, alpha ignored. 16-bit colour / grey inputs are REJECTED: image 0.19's DynamicImage has only
//                       8-bit variants and its decoder_to_image returns UnsupportedColor for them, so the reference cannot open
//                       such a file either.
// Decoding is checked against files written by an independent encoder (Pillow, adaptive filters 0-4, grey / RGB / RGBA / grey+alpha /
// palette / 1-bit / 16-bit): tests/golden/png/ + tests/test_host_plumbing.py. Non-interlaced PNG only (the crates decode Adam7 too;
// TUM RGB-D files are not interlaced).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace vors {
namespace png_io {

struct Image {
    uint32_t width = 0, height = 0;
    int bit_depth = 0, channels = 0, color_type = -1;
    std::vector<uint8_t> data;     // decoded, unfiltered scanlines (big-endian samples for 16-bit; packed samples below 8 bits)
    std::vector<uint8_t> palette;  // colour type 3: r g b triples
    size_t stride = 0;             // bytes per scanline
};

constexpr uint64_t MAX_PIXELS = 1ull << 28;  // the tracker's own limit (capi.cpp build_geom): refuse absurd headers before allocating

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline Image decode(const std::vector<uint8_t>& file) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file");
    Image img;
    std::vector<uint8_t> idat;
    int interlace = 0;
    bool have_ihdr = false;
    size_t p = 8;
    while (p + 12 <= file.size()) {
        const uint32_t len = be32(&file[p]);
        const char* type = reinterpret_cast<const char*>(&file[p + 4]);
        if ((uint64_t)p + 12 + len > file.size()) throw std::runtime_error("truncated PNG chunk");
        const uint8_t* d = &file[p + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) throw std::runtime_error("bad PNG IHDR length");
            img.width = be32(d);
            img.height = be32(d + 4);
            img.bit_depth = d[8];
            img.color_type = d[9];
            if (d[10] != 0 || d[11] != 0) throw std::runtime_error("unknown PNG compression / filter method");
            interlace = d[12];
            have_ihdr = true;
        } else if (!have_ihdr) {
            throw std::runtime_error("PNG does not start with IHDR");
        } else if (!std::memcmp(type, "PLTE", 4)) {
            if (len % 3 != 0 || len > 768) throw std::runtime_error("bad PNG palette");
            img.palette.assign(d, d + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), d, d + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        p += 12 + (size_t)len;
    }
    if (!have_ihdr) throw std::runtime_error("PNG without IHDR");
    if (img.width == 0 || img.height == 0 || (uint64_t)img.width * img.height > MAX_PIXELS)
        throw std::runtime_error("PNG dimensions out of range");
    if (interlace != 0) throw std::runtime_error("interlaced PNG not supported");
    const int bd = img.bit_depth;
    bool ok_depth = false;
    switch (img.color_type) {
        case 0: img.channels = 1; ok_depth = bd == 1 || bd == 2 || bd == 4 || bd == 8 || bd == 16; break;
        case 2: img.channels = 3; ok_depth = bd == 8 || bd == 16; break;
        case 3: img.channels = 1; ok_depth = bd == 1 || bd == 2 || bd == 4 || bd == 8; break;
        case 4: img.channels = 2; ok_depth = bd == 8 || bd == 16; break;
        case 6: img.channels = 4; ok_depth = bd == 8 || bd == 16; break;
        default: throw std::runtime_error("unsupported PNG colour type");
    }
    if (!ok_depth) throw std::runtime_error("PNG bit depth not allowed for its colour type");
    if (img.color_type == 3 && img.palette.empty()) throw std::runtime_error("palette PNG without PLTE");
    const size_t bits_pp = (size_t)img.channels * bd;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;  // filter distance in bytes (1 below 8 bits per pixel)
    const size_t stride = ((size_t)img.width * bits_pp + 7) / 8;
    img.stride = stride;
    std::vector<uint8_t> raw((stride + 1) * img.height);
    uLongf out_len = raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), idat.size()) != Z_OK || out_len != raw.size())
        throw std::runtime_error("PNG inflate failed");
    img.data.resize(stride * img.height);
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < img.height; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        if (ft > 4) throw std::runtime_error("bad PNG filter type");
        const uint8_t* in = &raw[y * (stride + 1) + 1];
        uint8_t* out = &img.data[y * stride];
        const uint8_t* up = y ? &img.data[(y - 1) * stride] : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? out[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                default: {
                    const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
            }
            out[i] = (uint8_t)(in[i] + pred);
        }
    }
    return img;
}

inline std::vector<uint8_t> read_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    return buf;
}

// helper.rs:13-36 -> (width, height, row-major u16)
inline void read_png_16bits(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint16_t>& out) {
    const Image img = decode(read_file(path));
    if (img.bit_depth != 16 || img.color_type != 0) throw std::runtime_error(path + ": expected a 16-bit grey PNG");
    w = img.width;
    h = img.height;
    out.resize((size_t)w * h);
    for (size_t i = 0; i < out.size(); ++i) out[i] = (uint16_t)(img.data[2 * i] << 8 | img.data[2 * i + 1]);  // BigEndian
}

// image 0.19 rgb_to_luma
inline uint8_t rgb_to_luma(uint8_t r, uint8_t g, uint8_t b) { return (uint8_t)(0.2126f * (float)r + 0.7152f * (float)g + 0.0722f * (float)b); }

// image::open(path).to_luma() -> row-major u8
inline void read_luma8(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& out) {
    const Image img = decode(read_file(path));
    if (img.bit_depth == 16)
        throw std::runtime_error(path + ": 16-bit colour/grey images are not supported (neither by the reference's image 0.19: UnsupportedColor)");
    w = img.width;
    h = img.height;
    out.resize((size_t)w * h);
    const int bd = img.bit_depth;
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* row = &img.data[(size_t)y * img.stride];
        uint8_t* o = &out[(size_t)y * w];
        for (uint32_t x = 0; x < w; ++x) {
            if (img.color_type == 0 || img.color_type == 3) {
                unsigned v;
                if (bd == 8) {
                    v = row[x];
                } else {  // packed, most significant bits first
                    const unsigned per = 8 / bd, sh = (per - 1 - x % per) * bd;
                    v = (row[x / per] >> sh) & ((1u << bd) - 1);
                }
                if (img.color_type == 3) {
                    if (3 * v + 2 >= img.palette.size()) throw std::runtime_error(path + ": palette index out of range");
                    o[x] = rgb_to_luma(img.palette[3 * v], img.palette[3 * v + 1], img.palette[3 * v + 2]);
                } else {
                    o[x] = (uint8_t)(bd == 8 ? v : v * 255u / ((1u << bd) - 1));
                }
            } else {
                const uint8_t* px = row + (size_t)x * img.channels;
                o[x] = img.channels == 2 ? px[0] : rgb_to_luma(px[0], px[1], px[2]);
            }
        }
    }
}

// Writers (synthetic TUM-format sequences for the tests): filter 0, zlib default compression.
inline void write_png(const std::string& path, uint32_t w, uint32_t h, int bit_depth, int color_type, const std::vector<uint8_t>& scan) {
    const size_t stride = scan.size() / h;
    std::vector<uint8_t> raw((stride + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw[y * (stride + 1)] = 0;
        std::memcpy(&raw[y * (stride + 1) + 1], &scan[y * stride], stride);
    }
    uLongf clen = compressBound(raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress(comp.data(), &clen, raw.data(), raw.size()) != Z_OK) throw std::runtime_error("deflate failed");
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    auto chunk = [&](const char* type, const uint8_t* d, uint32_t len) {
        uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)type[0], (uint8_t)type[1],
                          (uint8_t)type[2], (uint8_t)type[3]};
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(d, 1, len, f);
        uint32_t crc = crc32(0, hdr + 4, 4);
        if (len) crc = crc32(crc, d, len);
        const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
        std::fwrite(c, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16),
                        (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)bit_depth, (uint8_t)color_type, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    std::fclose(f);
}
inline void write_gray16(const std::string& path, uint32_t w, uint32_t h, const uint16_t* px) {
    std::vector<uint8_t> scan((size_t)w * h * 2);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        scan[2 * i] = (uint8_t)(px[i] >> 8);
        scan[2 * i + 1] = (uint8_t)px[i];
    }
    write_png(path, w, h, 16, 0, scan);
}
inline void write_gray8(const std::string& path, uint32_t w, uint32_t h, const uint8_t* px) {
    write_png(path, w, h, 8, 0, std::vector<uint8_t>(px, px + (size_t)w * h));
}
inline void write_rgb8(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgb) {
    write_png(path, w, h, 8, 2, std::vector<uint8_t>(rgb, rgb + (size_t)w * h * 3));
}

}  // namespace png_io
}  // namespace vors
