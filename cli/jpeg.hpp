// Baseline JPEG (SOF0 / SOF1, 8-bit, Huffman; grey or YCbCr with 1x1 / 2x1 / 2x2 chroma subsampling, restart
// intervals) for the CLIs: the reference reads its test photographs and writes its key frames as .jpg through OpenCV
// (modules/bgdehaze/main.py:16,19; modules/videostrip/src/main.cpp:294,377), and the build image has no codec headers.
//
// The decoder follows the published IJG / libjpeg(-turbo) algorithms step for step so that it returns the pixels
// cv::imread / Pillow return: "islow" integer IDCT (jidctint), "fancy" triangle chroma upsampling (jdsample
// h2v1 / h2v2_fancy_upsample), fixed-point YCbCr -> RGB (jdcolor).  tests/test_cli.py checks it bit for bit against
// Pillow's decode of the reference's four JPEG files.  The encoder is the baseline encoder of the same library:
// RGB -> YCbCr (jccolor), 2x2 chroma downsampling (jcsample), islow forward DCT (jfdctint), the Annex K tables scaled by
// quality (cv::imwrite's default: 95), the standard Huffman tables.  Pixels are BGR-interleaved (cv::Mat), or 1 channel.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace jpeg {

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- the slow-but-accurate integer DCT pair (CONST_BITS 13, PASS1_BITS 2) ------------------------------------------
enum { CB = 13, P1 = 2 };
static inline int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }
static inline int64_t descale64(int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; }
enum : int32_t { F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
                 F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172 };

// coef: 64 dequantised coefficients in natural order (|coef| <= 2^20, the decoder clamps: a valid stream stays far
// below); out: 64 samples 0..255.  64-bit temporaries: the same values as libjpeg's 32-bit ones on a valid stream, and no
// signed overflow on a corrupt one.
static inline void idct_islow(const int32_t *coef, uint8_t *out, int ostride)
{
    int64_t ws[64];
    for (int c = 0; c < 8; ++c) {
        const int32_t *in = coef + c;
        int64_t z2 = in[16], z3 = in[48];
        int64_t z1 = (z2 + z3) * F0541;
        int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        z2 = in[0]; z3 = in[32];
        int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        int64_t *w = ws + c;
        w[0] = descale64(tmp10 + tmp3, CB - P1);  w[56] = descale64(tmp10 - tmp3, CB - P1);
        w[8] = descale64(tmp11 + tmp2, CB - P1);  w[48] = descale64(tmp11 - tmp2, CB - P1);
        w[16] = descale64(tmp12 + tmp1, CB - P1); w[40] = descale64(tmp12 - tmp1, CB - P1);
        w[24] = descale64(tmp13 + tmp0, CB - P1); w[32] = descale64(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const int64_t *w = ws + r * 8;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F0541;
        int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        int64_t tmp0 = (w[0] + w[4]) * (1 << CB), tmp1 = (w[0] - w[4]) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        auto put = [&](int i, int64_t v) {
            const int64_t s = descale64(v, CB + P1 + 3) + 128;
            out[r * ostride + i] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
        };
        put(0, tmp10 + tmp3); put(7, tmp10 - tmp3); put(1, tmp11 + tmp2); put(6, tmp11 - tmp2);
        put(2, tmp12 + tmp1); put(5, tmp12 - tmp1); put(3, tmp13 + tmp0); put(4, tmp13 - tmp0);
    }
}

// d: 64 samples - 128 in natural order -> 64 coefficients scaled by 8 (jfdctint)
static inline void fdct_islow(int32_t *d)
{
    for (int r = 0; r < 8; ++r) {
        int32_t *p = d + r * 8;
        int32_t tmp0 = p[0] + p[7], tmp7 = p[0] - p[7], tmp1 = p[1] + p[6], tmp6 = p[1] - p[6];
        int32_t tmp2 = p[2] + p[5], tmp5 = p[2] - p[5], tmp3 = p[3] + p[4], tmp4 = p[3] - p[4];
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        p[0] = (tmp10 + tmp11) * (1 << P1); p[4] = (tmp10 - tmp11) * (1 << P1);
        int32_t z1 = (tmp12 + tmp13) * F0541;
        p[2] = descale(z1 + tmp13 * F0765, CB - P1); p[6] = descale(z1 + tmp12 * (-F1847), CB - P1);
        z1 = tmp4 + tmp7;
        int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
        const int32_t z5 = (z3 + z4) * F1175;
        tmp4 *= F0298; tmp5 *= F2053; tmp6 *= F3072; tmp7 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        p[7] = descale(tmp4 + z1 + z3, CB - P1); p[5] = descale(tmp5 + z2 + z4, CB - P1);
        p[3] = descale(tmp6 + z2 + z3, CB - P1); p[1] = descale(tmp7 + z1 + z4, CB - P1);
    }
    for (int c = 0; c < 8; ++c) {
        int32_t *p = d + c;
        int32_t tmp0 = p[0] + p[56], tmp7 = p[0] - p[56], tmp1 = p[8] + p[48], tmp6 = p[8] - p[48];
        int32_t tmp2 = p[16] + p[40], tmp5 = p[16] - p[40], tmp3 = p[24] + p[32], tmp4 = p[24] - p[32];
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        p[0] = descale(tmp10 + tmp11, P1); p[32] = descale(tmp10 - tmp11, P1);
        int32_t z1 = (tmp12 + tmp13) * F0541;
        p[16] = descale(z1 + tmp13 * F0765, CB + P1); p[48] = descale(z1 + tmp12 * (-F1847), CB + P1);
        z1 = tmp4 + tmp7;
        int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
        const int32_t z5 = (z3 + z4) * F1175;
        tmp4 *= F0298; tmp5 *= F2053; tmp6 *= F3072; tmp7 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        p[56] = descale(tmp4 + z1 + z3, CB + P1); p[40] = descale(tmp5 + z2 + z4, CB + P1);
        p[24] = descale(tmp6 + z2 + z3, CB + P1); p[8] = descale(tmp7 + z1 + z4, CB + P1);
    }
}

// ---- Huffman tables -----------------------------------------------------------------------------------------------
struct HuffTable {
    uint8_t bits[17] = {0}, vals[256] = {0};
    // decoding
    int32_t maxcode[18], valptr[17];
    uint16_t lookup[512];      // 9-bit look-ahead: (length << 8) | symbol, 0 = longer code
    // encoding
    uint16_t ecode[256];
    uint8_t esize[256];
    bool present = false;
    // false: the counts do not describe a prefix code (over-subscribed lengths, libjpeg's JERR_BAD_HUFF_TABLE)
    bool build()
    {
        present = false;
        uint8_t huffsize[257];
        uint16_t huffcode[257];
        int p = 0;
        for (int l = 1; l <= 16; ++l) for (int i = 0; i < bits[l]; ++i) { if (p >= 256) return false; huffsize[p++] = (uint8_t)l; }
        huffsize[p] = 0;
        const int n = p;
        int code = 0, si = huffsize[0];
        p = 0;
        while (huffsize[p]) {
            while (huffsize[p] == si) huffcode[p++] = (uint16_t)code++;
            if (code >= (1 << si)) return false;     // more codes of length si than the prefix tree has room for
            code <<= 1; si++;
        }
        p = 0;
        for (int l = 1; l <= 16; ++l) {
            if (bits[l]) { valptr[l] = p - (int)huffcode[p]; p += bits[l]; maxcode[l] = huffcode[p - 1]; }
            else maxcode[l] = -1;
        }
        maxcode[17] = 0xFFFFF;
        std::memset(lookup, 0, sizeof lookup);
        p = 0;
        for (int l = 1; l <= 9; ++l)
            for (int i = 0; i < bits[l]; ++i, ++p) {
                const int first = huffcode[p] << (9 - l);
                for (int k = 0; k < (1 << (9 - l)); ++k) lookup[first + k] = (uint16_t)((l << 8) | vals[p]);
            }
        std::memset(esize, 0, sizeof esize);
        for (int i = 0; i < n; ++i) { ecode[vals[i]] = huffcode[i]; esize[vals[i]] = huffsize[i]; }
        present = true;
        return true;
    }
};

// ---- decoder ---------------------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void fill()
    {
        while (nbits <= 24) {
            int c = 0;
            if (!hit_marker && p < end) {
                c = *p++;
                if (c == 0xFF) {
                    if (p < end && *p == 0) ++p;
                    else { hit_marker = true; --p; c = 0; }     // a marker: feed zeros, leave it for the caller
                }
            }
            acc |= (uint32_t)c << (24 - nbits);
            nbits += 8;
        }
    }
    int peek(int n) { if (nbits < n) fill(); return (int)(acc >> (32 - n)); }
    void skip(int n) { acc <<= n; nbits -= n; }
    int get(int n) { if (n == 0) return 0; const int v = peek(n); skip(n); return v; }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

static inline int huff_decode(BitReader &br, const HuffTable &h)
{
    const int look = br.peek(9);
    const uint16_t e = h.lookup[look];
    if (e) { br.skip(e >> 8); return e & 255; }
    int code = br.peek(16), l = 10;
    for (; l <= 16; ++l) if ((code >> (16 - l)) <= h.maxcode[l]) break;
    if (l > 16) return -1;
    br.skip(l);
    const int idx = h.valptr[l] + (code >> (16 - l));
    return (idx >= 0 && idx < 256) ? h.vals[idx] : -1;
}
static inline int32_t clamp_coef(int64_t v) { return (int32_t)(v > (1 << 20) ? (1 << 20) : (v < -(1 << 20) ? -(1 << 20) : v)); }
static inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int wblocks = 0, hblocks = 0;       // MCU-padded size in blocks
    int dw = 0, dh = 0;                 // downsampled size in samples
    std::vector<uint8_t> plane;         // wblocks*8 x hblocks*8
    int pred = 0;
};

inline void default_tables(HuffTable *dc, HuffTable *ac);

// Returns BGR (channels = 3) or grey (channels = 1, unless force_color) pixels.
inline bool decode(const uint8_t *buf, size_t len, int &rows, int &cols, int &channels, std::vector<uint8_t> &pix, bool force_color)
{
    if (len < 4 || buf[0] != 0xFF || buf[1] != 0xD8) return false;
    uint16_t qt[4][64] = {{0}};
    HuffTable dc[4], ac[4];
    default_tables(dc, ac);          // Motion-JPEG frames may omit DHT: the standard tables apply (overridden by any DHT)
    std::vector<Component> comp;
    int W = 0, H = 0, restart = 0, hmax = 1, vmax = 1;
    size_t pos = 2;
    bool have_sof = false;
    while (pos + 4 <= len) {
        if (buf[pos] != 0xFF) { ++pos; continue; }
        const int m = buf[pos + 1];
        pos += 2;
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0xFF) { if (m == 0xFF) --pos; continue; }
        if (m == 0xD9) break;
        if (pos + 2 > len) return false;
        const size_t L = ((size_t)buf[pos] << 8) | buf[pos + 1];
        if (L < 2 || pos + L > len) return false;
        const uint8_t *s = buf + pos + 2, *e = buf + pos + L;
        if (m == 0xDB) {
            while (s < e) {
                const int pq = s[0] >> 4, tq = s[0] & 15;
                ++s;
                if (tq > 3 || s + (pq ? 128 : 64) > e) return false;
                for (int i = 0; i < 64; ++i) { qt[tq][ZIGZAG[i]] = pq ? (uint16_t)((s[0] << 8) | s[1]) : s[0]; s += pq ? 2 : 1; }
            }
        } else if (m == 0xC4) {
            while (s + 17 <= e) {
                const int tc = s[0] >> 4, th = s[0] & 15;
                if (th > 3) return false;
                HuffTable &t = tc ? ac[th] : dc[th];
                int n = 0;
                t.bits[0] = 0;
                for (int i = 1; i <= 16; ++i) { t.bits[i] = s[i]; n += s[i]; }
                s += 17;
                if (n > 256 || s + n > e) return false;
                std::memcpy(t.vals, s, n);
                s += n;
                if (!t.build()) return false;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (L < 8 || s[0] != 8) return false;
            H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
            const int n = s[5];
            if ((n != 1 && n != 3) || W <= 0 || H <= 0 || L < (size_t)(8 + 3 * n)) return false;
            if (have_sof) return false;             // one frame per file
            comp.resize(n);
            for (int i = 0; i < n; ++i) {
                comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i] & 3;
                if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2) return false;
                hmax = comp[i].h > hmax ? comp[i].h : hmax; vmax = comp[i].v > vmax ? comp[i].v : vmax;
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return false;                   // progressive / lossless / arithmetic: not the baseline this decoder reads
        } else if (m == 0xDD) {
            if (L < 4) return false;
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof || L < 3) return false;
            const int ns = s[0];
            if (ns != (int)comp.size() || L < (size_t)(6 + 2 * ns)) return false;       // one interleaved scan (what baseline encoders write)
            for (int i = 0; i < ns; ++i)
                for (auto &c : comp) if (c.id == s[1 + 2 * i]) { c.td = s[2 + 2 * i] >> 4; c.ta = s[2 + 2 * i] & 15; }
            for (auto &c : comp) if (c.td > 3 || c.ta > 3) return false;
            pos += L;
            // ---- the entropy-coded segment
            const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (auto &c : comp) {
                c.wblocks = mcux * c.h; c.hblocks = mcuy * c.v;
                c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
                c.plane.assign((size_t)c.wblocks * 8 * c.hblocks * 8, 0);
                c.pred = 0;
                if (!dc[c.td].present || !ac[c.ta].present) return false;
            }
            BitReader br{buf + pos, buf + len};
            int count = 0;
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx) {
                    if (restart && count == restart) {
                        // expect RSTn at the byte position; resynchronise
                        br.reset();
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
                        if (br.p + 1 < br.end) br.p += 2;
                        for (auto &c : comp) c.pred = 0;
                        count = 0;
                    }
                    ++count;
                    for (auto &c : comp)
                        for (int by = 0; by < c.v; ++by)
                            for (int bx = 0; bx < c.h; ++bx) {
                                int32_t blk[64] = {0};
                                const int t = huff_decode(br, dc[c.td]);
                                if (t < 0 || t > 11) return false;
                                const int diff = t ? extend(br.get(t), t) : 0;
                                c.pred += diff;
                                if (c.pred > 32767 || c.pred < -32768) return false;     // a valid DC stays within 12 bits
                                blk[0] = clamp_coef((int64_t)c.pred * qt[c.tq][0]);
                                for (int k = 1; k < 64;) {
                                    const int rs = huff_decode(br, ac[c.ta]);
                                    if (rs < 0) return false;
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) return false;
                                    const int z = ZIGZAG[k];
                                    blk[z] = clamp_coef((int64_t)extend(br.get(sz), sz) * qt[c.tq][z]);
                                    ++k;
                                }
                                const int X = (mx * c.h + bx) * 8, Y = (my * c.v + by) * 8, stride = c.wblocks * 8;
                                idct_islow(blk, &c.plane[(size_t)Y * stride + X], stride);
                            }
                }
            break;
        }
        pos += L;
    }
    if (comp.empty() || comp[0].plane.empty()) return false;
    rows = H; cols = W;
    // ---- upsample (fancy, as libjpeg's default) + colour
    auto samp = [](const Component &c, int x, int y) -> int {     // edge-replicated access inside the downsampled plane
        x = x < 0 ? 0 : (x >= c.dw ? c.dw - 1 : x);
        y = y < 0 ? 0 : (y >= c.dh ? c.dh - 1 : y);
        return c.plane[(size_t)y * c.wblocks * 8 + x];
    };
    auto full = [&](const Component &c, std::vector<uint8_t> &o) {
        o.resize((size_t)W * H);
        const int hx = hmax / c.h, vx = vmax / c.v;
        if (hx == 1 && vx == 1) {
            for (int y = 0; y < H; ++y) std::memcpy(&o[(size_t)y * W], &c.plane[(size_t)y * c.wblocks * 8], W);
        } else if (hx == 2 && vx == 1) {                      // h2v1_fancy_upsample
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < c.dw; ++x) {
                    const int s0 = samp(c, x, y);
                    int a, b;
                    if (c.dw == 1) { a = b = s0; }
                    else if (x == 0) { a = s0; b = (s0 * 3 + samp(c, 1, y) + 2) >> 2; }
                    else if (x == c.dw - 1) { a = (s0 * 3 + samp(c, x - 1, y) + 1) >> 2; b = s0; }
                    else { a = (s0 * 3 + samp(c, x - 1, y) + 1) >> 2; b = (s0 * 3 + samp(c, x + 1, y) + 2) >> 2; }
                    if (2 * x < W) o[(size_t)y * W + 2 * x] = (uint8_t)a;
                    if (2 * x + 1 < W) o[(size_t)y * W + 2 * x + 1] = (uint8_t)b;
                }
        } else if (hx == 2 && vx == 2) {                      // h2v2_fancy_upsample
            for (int y = 0; y < H; ++y) {
                const int y0 = y >> 1, y1 = (y & 1) ? y0 + 1 : y0 - 1;         // nearer row, farther row
                for (int x = 0; x < c.dw; ++x) {
                    const int thisc = samp(c, x, y0) * 3 + samp(c, x, y1);
                    int a, b;
                    if (c.dw == 1) { a = (thisc * 4 + 8) >> 4; b = (thisc * 4 + 7) >> 4; }
                    else {
                        const int lastc = samp(c, x - 1, y0) * 3 + samp(c, x - 1, y1), nextc = samp(c, x + 1, y0) * 3 + samp(c, x + 1, y1);
                        a = x == 0 ? (thisc * 4 + 8) >> 4 : (thisc * 3 + lastc + 8) >> 4;
                        b = x == c.dw - 1 ? (thisc * 4 + 7) >> 4 : (thisc * 3 + nextc + 7) >> 4;
                    }
                    if (2 * x < W) o[(size_t)y * W + 2 * x] = (uint8_t)a;
                    if (2 * x + 1 < W) o[(size_t)y * W + 2 * x + 1] = (uint8_t)b;
                }
            }
        } else {                                              // 1x2: replicate rows (h1v2 has no fancy form in libjpeg 6b)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) o[(size_t)y * W + x] = (uint8_t)samp(c, x / hx, y / vx);
        }
    };
    std::vector<uint8_t> Y, Cb, Cr;
    full(comp[0], Y);
    if (comp.size() == 1) {
        channels = force_color ? 3 : 1;
        pix.resize((size_t)W * H * channels);
        for (size_t i = 0; i < (size_t)W * H; ++i)
            if (channels == 1) pix[i] = Y[i];
            else pix[3 * i] = pix[3 * i + 1] = pix[3 * i + 2] = Y[i];
        return true;
    }
    full(comp[1], Cb); full(comp[2], Cr);
    channels = 3;
    pix.resize((size_t)W * H * 3);
    // jdcolor.c build_ycc_rgb_table: SCALEBITS 16
    // built once, thread-safe (C++11 function-local static: `videostrip -g N` decodes from N worker threads)
    struct YccTables {
        int32_t crr[256], cbb[256], crg[256], cbg[256];
        YccTables() {
            for (int i = 0; i < 256; ++i) {
                const int32_t x = i - 128;
                crr[i] = (int32_t)((91881LL * x + 32768) >> 16);           // FIX(1.40200)
                cbb[i] = (int32_t)((116130LL * x + 32768) >> 16);          // FIX(1.77200)
                crg[i] = (int32_t)(-46802LL * x);                          // FIX(0.71414)
                cbg[i] = (int32_t)(-22554LL * x + 32768);                  // FIX(0.34414) + ONE_HALF
            }
        }
    };
    static const YccTables tab;
    const int32_t *crr = tab.crr, *cbb = tab.cbb, *crg = tab.crg, *cbg = tab.cbg;
    auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const int y = Y[i], cb = Cb[i], cr = Cr[i];
        pix[3 * i + 2] = clamp8(y + crr[cr]);
        pix[3 * i + 1] = clamp8(y + ((cbg[cb] + crg[cr]) >> 16));
        pix[3 * i + 0] = clamp8(y + cbb[cb]);
    }
    return true;
}

// ---- encoder (baseline, 4:2:0 for colour, standard tables, quality as cv::imwrite: default 95) ----------------------
static const uint8_t STD_LUM_Q[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                      69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55, 64,
                                      81, 104, 113, 92, 49, 64, 78,  87,  103, 121, 120, 101, 72, 92, 95,  98,  112, 100, 103, 99};
static const uint8_t STD_CHR_Q[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                      99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const uint8_t DC_LUM_BITS[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_CHR_BITS[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_LUM_BITS[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t AC_LUM_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
static const uint8_t AC_CHR_BITS[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t AC_CHR_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};

inline void std_table(HuffTable &t, const uint8_t *bits, const uint8_t *vals, int n)
{
    std::memcpy(t.bits, bits, 17);
    std::memcpy(t.vals, vals, n);
    t.build();
}

inline void default_tables(HuffTable *dc, HuffTable *ac)
{
    std_table(dc[0], DC_LUM_BITS, DC_VALS, 12); std_table(ac[0], AC_LUM_BITS, AC_LUM_VALS, 162);
    std_table(dc[1], DC_CHR_BITS, DC_VALS, 12); std_table(ac[1], AC_CHR_BITS, AC_CHR_VALS, 162);
}

struct BitWriter {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    void put(uint32_t code, int size)
    {
        acc = (acc << size) | (code & ((1u << size) - 1));
        n += size;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);
            n -= 8;
        }
    }
    void flush() { if (n) put(0x7F, 8 - n); }        // pad with ones
};

inline bool encode(const uint8_t *pix, int rows, int cols, int channels, int quality, std::vector<uint8_t> &out)
{
    if (rows <= 0 || cols <= 0 || (channels != 1 && channels != 3)) return false;
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t q[2][64];
    for (int i = 0; i < 64; ++i) {
        int a = (STD_LUM_Q[i] * scale + 50) / 100, b = (STD_CHR_Q[i] * scale + 50) / 100;
        q[0][i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
        q[1][i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
    HuffTable dcl, dcc, acl, acc_;
    std_table(dcl, DC_LUM_BITS, DC_VALS, 12); std_table(dcc, DC_CHR_BITS, DC_VALS, 12);
    std_table(acl, AC_LUM_BITS, AC_LUM_VALS, 162); std_table(acc_, AC_CHR_BITS, AC_CHR_VALS, 162);
    const int nc = channels == 3 ? 3 : 1, hs = nc == 3 ? 2 : 1;
    const int mcu = 8 * hs, mcux = (cols + mcu - 1) / mcu, mcuy = (rows + mcu - 1) / mcu;
    const int PW = mcux * mcu, PH = mcuy * mcu;
    // colour conversion (jccolor.c rgb_ycc_convert) into edge-replicated planes
    std::vector<uint8_t> Y((size_t)PW * PH), Cb, Cr;
    if (nc == 3) { Cb.resize((size_t)PW * PH); Cr.resize((size_t)PW * PH); }
    for (int y = 0; y < PH; ++y)
        for (int x = 0; x < PW; ++x) {
            const uint8_t *p = pix + ((size_t)(y < rows ? y : rows - 1) * cols + (x < cols ? x : cols - 1)) * channels;
            if (nc == 1) { Y[(size_t)y * PW + x] = p[0]; continue; }
            const int b = p[0], g = p[1], r = p[2];
            Y[(size_t)y * PW + x] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
            Cb[(size_t)y * PW + x] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
            Cr[(size_t)y * PW + x] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
        }
    // h2v2_downsample: 2x2 box with the alternating 1, 2 bias
    std::vector<uint8_t> cb2, cr2;
    const int CW = PW / 2, CH = PH / 2;
    if (nc == 3) {
        cb2.resize((size_t)CW * CH); cr2.resize((size_t)CW * CH);
        for (int y = 0; y < CH; ++y)
            for (int x = 0; x < CW; ++x) {
                const int bias = 1 + (x & 1);
                auto box = [&](const std::vector<uint8_t> &P) {
                    return (uint8_t)((P[(size_t)(2 * y) * PW + 2 * x] + P[(size_t)(2 * y) * PW + 2 * x + 1] + P[(size_t)(2 * y + 1) * PW + 2 * x] +
                                      P[(size_t)(2 * y + 1) * PW + 2 * x + 1] + bias) >> 2);
                };
                cb2[(size_t)y * CW + x] = box(Cb); cr2[(size_t)y * CW + x] = box(Cr);
            }
    }
    out.clear();
    auto w16 = [&](int v) { out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v); };
    out.push_back(0xFF); out.push_back(0xD8);
    static const uint8_t jfif[16] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1};
    out.insert(out.end(), jfif, jfif + 16); out.push_back(0); out.push_back(0);
    for (int t = 0; t < (nc == 3 ? 2 : 1); ++t) {
        out.push_back(0xFF); out.push_back(0xDB); w16(67); out.push_back((uint8_t)t);
        for (int i = 0; i < 64; ++i) out.push_back(q[t][ZIGZAG[i]]);
    }
    out.push_back(0xFF); out.push_back(0xC0); w16(8 + 3 * nc); out.push_back(8); w16(rows); w16(cols); out.push_back((uint8_t)nc);
    for (int i = 0; i < nc; ++i) { out.push_back((uint8_t)(i + 1)); out.push_back((uint8_t)(i == 0 ? (hs << 4 | hs) : 0x11)); out.push_back((uint8_t)(i ? 1 : 0)); }
    auto dht = [&](int cls_id, const uint8_t *bits, const uint8_t *vals, int n) {
        out.push_back(0xFF); out.push_back(0xC4); w16(19 + n); out.push_back((uint8_t)cls_id);
        out.insert(out.end(), bits + 1, bits + 17); out.insert(out.end(), vals, vals + n);
    };
    dht(0x00, DC_LUM_BITS, DC_VALS, 12); dht(0x10, AC_LUM_BITS, AC_LUM_VALS, 162);
    if (nc == 3) { dht(0x01, DC_CHR_BITS, DC_VALS, 12); dht(0x11, AC_CHR_BITS, AC_CHR_VALS, 162); }
    out.push_back(0xFF); out.push_back(0xDA); w16(6 + 2 * nc); out.push_back((uint8_t)nc);
    for (int i = 0; i < nc; ++i) { out.push_back((uint8_t)(i + 1)); out.push_back((uint8_t)(i ? 0x11 : 0x00)); }
    out.push_back(0); out.push_back(63); out.push_back(0);
    BitWriter bw{out};
    int pred[3] = {0, 0, 0};
    auto block = [&](const uint8_t *P, int stride, const uint8_t *qt, const HuffTable &D, const HuffTable &A, int &pr) {
        int32_t d[64];
        for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) d[y * 8 + x] = (int32_t)P[(size_t)y * stride + x] - 128;
        fdct_islow(d);
        int zz[64];
        for (int i = 0; i < 64; ++i) {
            const int z = ZIGZAG[i];
            const int32_t qv = (int32_t)qt[z] << 3;
            int32_t t = d[z];
            if (t < 0) { t = -t; t += qv >> 1; t = t >= qv ? t / qv : 0; t = -t; }
            else { t += qv >> 1; t = t >= qv ? t / qv : 0; }
            zz[i] = t;
        }
        int diff = zz[0] - pr;
        pr = zz[0];
        int t = diff < 0 ? -diff : diff, nb = 0;
        while (t) { nb++; t >>= 1; }
        bw.put(D.ecode[nb], D.esize[nb]);
        if (nb) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), nb);
        int run = 0;
        for (int k = 1; k < 64; ++k) {
            int v = zz[k];
            if (v == 0) { run++; continue; }
            while (run > 15) { bw.put(A.ecode[0xF0], A.esize[0xF0]); run -= 16; }
            int a = v < 0 ? -v : v, n2 = 0;
            while (a) { n2++; a >>= 1; }
            const int sym = (run << 4) | n2;
            bw.put(A.ecode[sym], A.esize[sym]);
            bw.put((uint32_t)(v < 0 ? v - 1 : v), n2);
            run = 0;
        }
        if (run) bw.put(A.ecode[0], A.esize[0]);
    };
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            for (int by = 0; by < hs; ++by)
                for (int bx = 0; bx < hs; ++bx)
                    block(&Y[(size_t)(my * mcu + by * 8) * PW + mx * mcu + bx * 8], PW, q[0], dcl, acl, pred[0]);
            if (nc == 3) {
                block(&cb2[(size_t)(my * 8) * CW + mx * 8], CW, q[1], dcc, acc_, pred[1]);
                block(&cr2[(size_t)(my * 8) * CW + mx * 8], CW, q[1], dcc, acc_, pred[2]);
            }
        }
    bw.flush();
    out.push_back(0xFF); out.push_back(0xD9);
    return true;
}

}  // namespace jpeg
