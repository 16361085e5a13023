// Minimal PNG reader for the native host layer: what cv::imread(path, CV_LOAD_IMAGE_COLOR) delivers for the reference's inputs
// (reference cpp_code/src/data_io.cpp:48-72, DataIO::importImages) -- an 8-bit, 3-channel, BGR, row-major image.
// Supported: gray (1, 2, 4, 8, 16 bit), gray + alpha, RGB, RGBA and palette (1, 2, 4, 8 bit) PNGs, 16-bit samples reduced to
// their high byte (as libpng's strip-16 does for imread), sub-byte gray expanded by bit replication, non-interlaced.  Adam7 files are reported as unsupported (JPEG: esfm_jpeg.hpp) -- image decoding is host
// I/O outside the hot path, this reader exists so that the C++ executable needs no Python.
// Needs zlib (-lz) for the inflate step; everything else (chunk walk, CRC check, the five scanline filters) is here.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace p3dv {
namespace png {

inline uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Decodes `path` into bgr (rows x cols x 3).  Returns an empty string on success, otherwise what went wrong.
inline std::string read_bgr(const std::string &path, int &rows, int &cols, std::vector<uint8_t> &bgr)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    std::vector<uint8_t> file;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return path + " is not a PNG file";
    uint32_t width = 0, height = 0;
    int depth = 0, color = -1, interlace = 0;
    std::vector<uint8_t> idat, palette;
    size_t pos = 8;
    bool have_end = false;
    while (pos + 12 <= file.size() && !have_end) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
        if (pos + 12 + (size_t)len > file.size()) return path + ": truncated chunk";
        if (be32(data + len) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4)) return path + ": chunk CRC mismatch";
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return path + ": bad IHDR";
            width = be32(data); height = be32(data + 4); depth = data[8]; color = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return path + ": unknown compression / filter method";
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            have_end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (color < 0 || width == 0 || height == 0) return path + ": no image header";
    // bound the header before anything is sized from it: (stride + 1) * height must not overflow and a crafted IHDR must come back
    // as an error string, not as std::length_error / bad_alloc (65 536 x 65 536 at most, 2^28 pixels = 1 GiB of RGBA8 at most)
    if (width > (1u << 16) || height > (1u << 16) || (uint64_t)width * (uint64_t)height > ((uint64_t)1 << 28))
        return path + ": image dimensions out of range";
    if (interlace != 0) return path + ": Adam7-interlaced PNGs are not supported";
    int channels;
    switch (color) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return path + ": unknown colour type";
    }
    const bool sub_byte = depth == 1 || depth == 2 || depth == 4;
    if (sub_byte ? !(color == 0 || color == 3) : !(depth == 8 || (depth == 16 && color != 3))) return path + ": unsupported bit depth for this colour type";
    // filters work on whole bytes: the "pixel" distance is one byte for sub-byte samples
    const size_t bps = sub_byte ? 1 : (size_t)depth / 8, bpp = bps * (size_t)channels;
    const size_t stride = sub_byte ? ((size_t)width * (size_t)depth + 7) / 8 : bpp * width;
    std::vector<uint8_t> raw;
    try { raw.resize((stride + 1) * (size_t)height); } catch (const std::exception &) { return path + ": out of memory"; }
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return path + ": inflate failed";
    // undo the scanline filters in place (filter byte + stride bytes per row)
    std::vector<uint8_t> prev(stride, 0);
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t *row = &raw[(stride + 1) * (size_t)y];
        const int ft = row[0];
        uint8_t *cur = row + 1;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int add;
            switch (ft) {
                case 0: add = 0; break;
                case 1: add = a; break;
                case 2: add = b; break;
                case 3: add = (a + b) >> 1; break;
                case 4: add = paeth(a, b, c); break;
                default: return path + ": unknown scanline filter";
            }
            cur[i] = (uint8_t)(cur[i] + add);
        }
        std::memcpy(prev.data(), cur, stride);
    }
    rows = (int)height; cols = (int)width;
    try { bgr.resize((size_t)rows * cols * 3); } catch (const std::exception &) { return path + ": out of memory"; }
    for (int y = 0; y < rows; ++y) {
        const uint8_t *cur = &raw[(stride + 1) * (size_t)y + 1];
        uint8_t *dst = &bgr[(size_t)y * cols * 3];
        for (int x = 0; x < cols; ++x) {
            // whole-byte samples: pixel x starts at byte x * bpp (16-bit: big endian, the high byte comes first);
            // sub-byte samples are addressed by bit, leftmost pixel in the high-order bits -- p must not be formed from x * bpp there
            // (it would run up to width - stride bytes past the row)
            const uint8_t *p = cur;
            uint8_t sample;
            if (sub_byte) {
                const size_t bit = (size_t)x * (size_t)depth;
                sample = (uint8_t)((cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1));
            } else {
                p = cur + (size_t)x * bpp;
                sample = p[0];
            }
            uint8_t r, g, b;
            if (color == 0 || color == 4) { r = g = b = sub_byte ? (uint8_t)(sample * (255 / ((1 << depth) - 1))) : sample; }
            else if (color == 3) {
                const size_t e = (size_t)sample * 3;
                if (e + 3 > palette.size()) return path + ": palette index out of range";
                r = palette[e]; g = palette[e + 1]; b = palette[e + 2];
            } else { r = p[0]; g = p[bps]; b = p[2 * bps]; }
            dst[3 * x] = b; dst[3 * x + 1] = g; dst[3 * x + 2] = r;      // alpha, if any, is dropped as IMREAD_COLOR does
        }
    }
    return std::string();
}

}  // namespace png
}  // namespace p3dv
