// Baseline JPEG reader for the native host layer: what cv::imread(path, CV_LOAD_IMAGE_COLOR) delivers for the reference's JPEG
// inputs (reference cpp_code/src/data_io.cpp:48-72, DataIO::importImages; the COLMAP data sets of script/run_gerrardhall.sh,
// run_personhall.sh, run_southbuilding.sh are .JPG) -- an 8-bit, 3-channel, BGR, row-major image.
// cv::imread decodes through libjpeg(-turbo) with its defaults; this file restates those from the published algorithms
// [upstream: IJG libjpeg jdhuff.c, jidctint.c, jdsample.c, jdcolor.c] -- parity unpinned, like the rest of the host layer:
//   * sequential DCT, Huffman, 8-bit samples (SOF0 / SOF1), 1 or 3 components, restart intervals;
//   * the "islow" integer inverse DCT (13-bit constants, two passes, the exact shifts and rounding of jidctint.c);
//   * "fancy" (triangle-filter) chroma upsampling for 2:1 horizontal (h2v1) and 2:1 x 2:1 (h2v2) subsampling, replication else;
//   * YCbCr -> RGB with libjpeg's 16-bit fixed-point tables.
// Progressive (SOF2), arithmetic-coded, 12-bit, CMYK files are reported as unsupported; EXIF orientation is not applied.
// Image decoding is host I/O outside the hot path.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace p3dv {
namespace jpeg {

struct Huff {
    // canonical code tables of one DHT: for code length l (1..16) the smallest code, the largest code and the index of its first value
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int bw = 0, bh = 0;            // width / height in 8 x 8 blocks, padded to whole MCUs
    int dw = 0, dh = 0;            // downsampled size in samples (what the upsampler sees)
    std::vector<uint8_t> pix;      // bh*8 rows of bw*8 samples
    int pred = 0;
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    int get_bit()
    {
        if (nbits == 0) {
            uint8_t b = 0;
            if (p < end && !hit_marker) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) ++p;                       // stuffed zero
                    else { hit_marker = true; --p; b = 0; }               // a marker: feed zeros (libjpeg does the same and warns)
                }
            }
            acc = b; nbits = 8;
        }
        --nbits;
        return (acc >> nbits) & 1;
    }
    int get_bits(int n) { int v = 0; while (n--) v = (v << 1) | get_bit(); return v; }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader &br, const Huff &h)
{
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.get_bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

// jidctint.c: 8 x 8 inverse DCT of dequantised coefficients (natural order) to samples, bit for bit
inline void idct_islow(const int *coef, uint8_t *out, int pitch)
{
    constexpr int CB = 13, P1 = 2;
    constexpr long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                   F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    long ws[64];
    auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
    for (int c = 0; c < 8; ++c) {
        const int *in = coef + c;
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F0_541;
        long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        z2 = in[0]; z3 = in[32];
        long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[c] = descale(tmp10 + tmp3, CB - P1); ws[56 + c] = descale(tmp10 - tmp3, CB - P1);
        ws[8 + c] = descale(tmp11 + tmp2, CB - P1); ws[48 + c] = descale(tmp11 - tmp2, CB - P1);
        ws[16 + c] = descale(tmp12 + tmp1, CB - P1); ws[40 + c] = descale(tmp12 - tmp1, CB - P1);
        ws[24 + c] = descale(tmp13 + tmp0, CB - P1); ws[32 + c] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const long *w = ws + 8 * r;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F0_541;
        long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        long tmp0 = (w[0] + w[4]) << CB, tmp1 = (w[0] - w[4]) << CB;
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const long o[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
        for (int c = 0; c < 8; ++c) {
            long v = descale(o[c], CB + P1 + 3) + 128;
            out[r * pitch + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
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
    if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8) return path + " is not a JPEG file";
    static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    int qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comp;
    int width = 0, height = 0, restart = 0, hmax = 1, vmax = 1;
    bool have_sof = false, adobe = false;
    int adobe_transform = -1;
    size_t pos = 2;
    try {
        while (pos + 4 <= file.size()) {
            if (file[pos] != 0xFF) return path + ": marker expected";
            const int m = file[pos + 1];
            if (m == 0xFF) { ++pos; continue; }
            if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { pos += 2; continue; }
            if (m == 0xD9) break;
            const size_t len = ((size_t)file[pos + 2] << 8) | file[pos + 3];
            if (len < 2 || pos + 2 + len > file.size()) return path + ": truncated segment";
            const uint8_t *d = &file[pos + 4];
            const size_t dl = len - 2;
            if (m == 0xDB) {                                      // DQT
                size_t o = 0;
                while (o < dl) {
                    const int pq = d[o] >> 4, tq = d[o] & 15;
                    if (tq > 3 || o + 1 + (size_t)64 * (pq ? 2 : 1) > dl) return path + ": bad DQT";
                    ++o;
                    for (int k = 0; k < 64; ++k) { qt[tq][zz[k]] = pq ? ((d[o] << 8) | d[o + 1]) : d[o]; o += pq ? 2 : 1; }
                    have_qt[tq] = true;
                }
            } else if (m == 0xC4) {                               // DHT
                size_t o = 0;
                while (o < dl) {
                    if (o + 17 > dl) return path + ": bad DHT";
                    const int tc = d[o] >> 4, th = d[o] & 15;
                    if (tc > 1 || th > 3) return path + ": bad DHT";
                    Huff &h = tc ? ac[th] : dc[th];
                    int count = 0, code = 0, k = 0;
                    for (int l = 1; l <= 16; ++l) count += d[o + l];
                    if (count > 256 || o + 17 + (size_t)count > dl) return path + ": bad DHT";
                    for (int l = 1; l <= 16; ++l) {
                        const int nl = d[o + l];
                        h.valptr[l] = k; h.mincode[l] = code;
                        h.maxcode[l] = nl ? code + nl - 1 : -1;
                        code = (code + nl) << 1; k += nl;
                    }
                    std::memcpy(h.vals, d + o + 17, (size_t)count);
                    h.present = true;
                    o += 17 + (size_t)count;
                }
            } else if (m == 0xC0 || m == 0xC1) {                  // SOF0 / SOF1
                if (dl < 6) return path + ": bad SOF";
                if (d[0] != 8) return path + ": only 8-bit JPEG samples are supported";
                height = (d[1] << 8) | d[2]; width = (d[3] << 8) | d[4];
                const int nc = d[5];
                if ((nc != 1 && nc != 3) || dl < 6 + (size_t)3 * nc) return path + ": only 1- or 3-component JPEGs are supported";
                if (width <= 0 || height <= 0 || width > 65535 || height > 65535 || (uint64_t)width * height > (1ull << 28)) return path + ": implausible JPEG dimensions";
                comp.assign((size_t)nc, Component());
                hmax = vmax = 1;                                   // (a second SOF starts over)
                for (int c = 0; c < nc; ++c) {
                    comp[c].id = d[6 + 3 * c]; comp[c].h = d[7 + 3 * c] >> 4; comp[c].v = d[7 + 3 * c] & 15; comp[c].tq = d[8 + 3 * c];
                    if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) return path + ": unsupported sampling factors";
                    hmax = comp[c].h > hmax ? comp[c].h : hmax; vmax = comp[c].v > vmax ? comp[c].v : vmax;
                }
                have_sof = true;
            } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
                return path + ": progressive / lossless / arithmetic-coded JPEG is not supported (baseline only)";
            } else if (m == 0xDD) {                               // DRI
                if (dl >= 2) restart = (d[0] << 8) | d[1];
            } else if (m == 0xEE && dl >= 12 && !std::memcmp(d, "Adobe", 5)) {
                adobe = true; adobe_transform = d[11];
            } else if (m == 0xDA) {                               // SOS: the one scan of a baseline file
                if (!have_sof) return path + ": SOS before SOF";
                const int ns = d[0];
                if (ns != (int)comp.size() || dl < 1 + (size_t)2 * ns + 3) return path + ": multi-scan baseline JPEG is not supported";
                for (int s = 0; s < ns; ++s) {
                    const int cid = d[1 + 2 * s];
                    bool found = false;
                    for (auto &c : comp) if (c.id == cid) { c.td = d[2 + 2 * s] >> 4; c.ta = d[2 + 2 * s] & 15; found = true; }
                    if (!found) return path + ": scan names an unknown component";
                }
                for (auto &c : comp) {
                    if (c.td > 3 || c.ta > 3 || !dc[c.td].present || !ac[c.ta].present || !have_qt[c.tq]) return path + ": missing Huffman or quantisation table";
                    if (comp.size() == 1) { c.h = c.v = 1; }
                }
                if (comp.size() == 1) hmax = vmax = 1;
                const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
                for (auto &c : comp) {
                    c.bw = mcux * c.h; c.bh = mcuy * c.v;
                    c.dw = (width * c.h + hmax - 1) / hmax; c.dh = (height * c.v + vmax - 1) / vmax;
                    c.pix.assign((size_t)c.bw * 8 * c.bh * 8, 0);
                    c.pred = 0;
                }
                BitReader br{&file[pos + 2 + len], file.data() + file.size()};
                int coef[64];
                int since_restart = 0, next_rst = 0;
                for (int my = 0; my < mcuy; ++my)
                    for (int mx = 0; mx < mcux; ++mx) {
                        if (restart && since_restart == restart) {
                            // byte-align, expect RSTn
                            br.reset();
                            const uint8_t *q = br.p;
                            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                            if (q + 1 >= br.end) return path + ": restart marker missing";
                            (void)next_rst;
                            br.p = q + 2;
                            for (auto &c : comp) c.pred = 0;
                            since_restart = 0;
                        }
                        for (auto &c : comp)
                            for (int by = 0; by < c.v; ++by)
                                for (int bx = 0; bx < c.h; ++bx) {
                                    std::memset(coef, 0, sizeof(coef));
                                    const int t = decode_symbol(br, dc[c.td]);
                                    if (t < 0 || t > 15) return path + ": bad DC code";
                                    const int diff = t ? extend(br.get_bits(t), t) : 0;
                                    // (a crafted stream can run the predictor or a dequantised coefficient out of int: libjpeg bounds
                                    // coefficients to 16 bits after dequantisation; anything beyond 2^15 * 255 is rejected here)
                                    const long long pred = (long long)c.pred + diff;
                                    if (pred < -(1 << 20) || pred > (1 << 20)) return path + ": DC predictor out of range";
                                    c.pred = (int)pred;
                                    const long long dcq = pred * (long long)qt[c.tq][0];
                                    if (dcq < -8355840LL || dcq > 8355840LL) return path + ": coefficient out of range";
                                    coef[0] = (int)dcq;
                                    for (int k = 1; k < 64;) {
                                        const int rs = decode_symbol(br, ac[c.ta]);
                                        if (rs < 0) return path + ": bad AC code";
                                        const int r = rs >> 4, s = rs & 15;
                                        if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                                        k += r;
                                        if (k > 63) return path + ": AC run past the block";
                                        const long long acq = (long long)extend(br.get_bits(s), s) * (long long)qt[c.tq][zz[k]];
                                        if (acq < -8355840LL || acq > 8355840LL) return path + ": coefficient out of range";
                                        coef[zz[k]] = (int)acq;
                                        ++k;
                                    }
                                    const int px = (mx * c.h + bx) * 8, py = (my * c.v + by) * 8;
                                    idct_islow(coef, &c.pix[(size_t)py * c.bw * 8 + px], c.bw * 8);
                                }
                        ++since_restart;
                    }
                break;            // baseline: one scan holds everything
            }
            pos += 2 + len;
        }
        if (!have_sof || comp.empty() || comp[0].pix.empty()) return path + ": no image data";
        rows = height; cols = width;
        bgr.assign((size_t)rows * cols * 3, 0);
        // upsample every component to full resolution (libjpeg's "fancy" triangle filters for 2:1, plain copies for 1:1)
        std::vector<std::vector<uint8_t>> full(comp.size());
        for (size_t ci = 0; ci < comp.size(); ++ci) {
            const Component &c = comp[ci];
            const int pitch = c.bw * 8;
            std::vector<uint8_t> &o = full[ci];
            o.assign((size_t)rows * cols, 0);
            const int hs = hmax / c.h, vs = vmax / c.v;
            auto S = [&](int y, int x) -> int {            // replicate the last real row / column (what libjpeg's edge handling amounts to)
                y = y < 0 ? 0 : (y >= c.dh ? c.dh - 1 : y);
                x = x < 0 ? 0 : (x >= c.dw ? c.dw - 1 : x);
                return c.pix[(size_t)y * pitch + x];
            };
            for (int y = 0; y < rows; ++y)
                for (int x = 0; x < cols; ++x) {
                    int v;
                    if (hs == 1 && vs == 1) v = S(y, x);
                    else if (hs == 2 && vs == 1) {         // h2v1_fancy_upsample: 3/4 nearer + 1/4 farther, rounding 1 (even) / 2 (odd)
                        const int i = x >> 1;
                        if (c.dw == 1) v = S(y, 0);
                        else if (x == 0) v = S(y, 0);
                        else if (x == 2 * c.dw - 1) v = S(y, c.dw - 1);
                        else v = (x & 1) ? (3 * S(y, i) + S(y, i + 1) + 2) >> 2 : (3 * S(y, i) + S(y, i - 1) + 1) >> 2;
                    } else if (hs == 2 && vs == 2) {       // h2v2_fancy_upsample: column sums 3 * nearer row + farther row, then 3:1 across, >> 4
                        const int j = y >> 1, jf = (y & 1) ? j + 1 : j - 1, i = x >> 1;
                        auto colsum = [&](int xx) { return 3 * S(j, xx) + S(jf, xx); };
                        const int t = colsum(i);
                        if (c.dw == 1) v = (t * 4 + 8) >> 4;
                        else if (x == 0) v = (t * 4 + 8) >> 4;
                        else if (x == 2 * c.dw - 1) v = (t * 4 + 7) >> 4;
                        else v = (x & 1) ? (t * 3 + colsum(i + 1) + 7) >> 4 : (t * 3 + colsum(i - 1) + 8) >> 4;
                    } else v = S(y / vs, x / hs);          // h1v2: replication (libjpeg-turbo has a triangle filter for this rare layout too: such files are NOT bit-identical to cv::imread)
                    o[(size_t)y * cols + x] = (uint8_t)v;
                }
        }
        if (comp.size() == 1) {
            for (size_t i = 0; i < (size_t)rows * cols; ++i) bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = full[0][i];
        } else {
            const bool rgb_direct = adobe && adobe_transform == 0;      // Adobe marker, transform 0: the components ARE R, G, B
            int cr_r[256], cb_b[256];
            long cr_g[256], cb_g[256];
            for (int i = 0; i < 256; ++i) {                              // jdcolor.c build_ycc_rgb_table, SCALEBITS = 16
                const long x = i - 128;
                cr_r[i] = (int)((91881L * x + 32768) >> 16);
                cb_b[i] = (int)((116130L * x + 32768) >> 16);
                cr_g[i] = -46802L * x;
                cb_g[i] = -22554L * x + 32768;
            }
            auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
            for (size_t i = 0; i < (size_t)rows * cols; ++i) {
                const int y = full[0][i], cb = full[1][i], cr = full[2][i];
                if (rgb_direct) { bgr[3 * i] = (uint8_t)cr; bgr[3 * i + 1] = (uint8_t)cb; bgr[3 * i + 2] = (uint8_t)y; continue; }
                bgr[3 * i + 2] = clamp(y + cr_r[cr]);
                bgr[3 * i + 1] = clamp(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
                bgr[3 * i] = clamp(y + cb_b[cb]);
            }
        }
    } catch (const std::exception &e) {
        return path + ": " + e.what();
    }
    return std::string();
}

}  // namespace jpeg
}  // namespace p3dv

#include "esfm_png.hpp"

namespace p3dv {
// cv::imread's dispatch on the file signature: PNG or JPEG into an 8-bit BGR image; empty string on success
inline std::string read_image_bgr(const std::string &path, int &rows, int &cols, std::vector<uint8_t> &bgr)
{
    uint8_t sig[2] = {0, 0};
    if (FILE *f = std::fopen(path.c_str(), "rb")) { const size_t got = std::fread(sig, 1, 2, f); (void)got; std::fclose(f); }
    else return "cannot open " + path;
    if (sig[0] == 0xFF && sig[1] == 0xD8) return jpeg::read_bgr(path, rows, cols, bgr);
    return png::read_bgr(path, rows, cols, bgr);
}
}  // namespace p3dv
