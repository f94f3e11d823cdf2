// Minimal image I/O for the CLIs: 8-bit PNG (non-interlaced grey / RGB / RGBA, via zlib), binary PPM / PGM and
// baseline JPEG (cli/jpeg.hpp: libjpeg's algorithms restated, no codec library in the build image).  Pixels are returned
// BGR-interleaved, as cv::imread(CV_LOAD_IMAGE_COLOR) does (modules/histretch/src/histretch.cpp:158).
#pragma once
#include <zlib.h>
#include <cctype>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "jpeg.hpp"

namespace imgio {

struct Image {
    int rows = 0, cols = 0, channels = 0;        // channels: 1 or 3 (BGR)
    std::vector<uint8_t> data;
    bool empty() const { return data.empty(); }
};

inline bool ends_with(const std::string &s, const char *suf)
{
    const size_t n = std::strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i) if (std::tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
    return true;
}

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline bool read_file(const std::string &path, std::vector<uint8_t> &buf)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && std::fread(buf.data(), 1, buf.size(), f) == buf.size();
    std::fclose(f);
    return ok;
}

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

inline bool read_png(const std::vector<uint8_t> &buf, Image &img, bool force_color)
{
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (buf.size() < 33 || std::memcmp(buf.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> z;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const char *type = (const char *)&buf[pos + 4];
        if (pos + 12 + len > buf.size()) return false;
        const uint8_t *d = &buf[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) { W = be32(d); H = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (!std::memcmp(type, "IDAT", 4)) z.insert(z.end(), d, d + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (depth != 8 || interlace != 0 || W == 0 || H == 0) return false;
    const int spp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!spp) return false;                                            // palette images are not supported
    const size_t stride = (size_t)W * spp;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, z.data(), (uLong)z.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<uint8_t> px(stride * H);
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t ft = raw[y * (stride + 1)];
        const uint8_t *in = &raw[y * (stride + 1) + 1];
        uint8_t *out = &px[y * stride];
        const uint8_t *up = y ? &px[(y - 1) * stride] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)spp ? out[i - spp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)spp) ? up[i - spp] : 0;
            int v = in[i];
            switch (ft) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
            }
            out[i] = (uint8_t)v;
        }
    }
    const bool grey = spp <= 2;
    img.rows = (int)H; img.cols = (int)W; img.channels = (grey && !force_color) ? 1 : 3;
    img.data.resize((size_t)H * W * img.channels);
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        const uint8_t *p = &px[i * spp];
        if (img.channels == 1) img.data[i] = p[0];
        else if (grey) { img.data[3 * i] = img.data[3 * i + 1] = img.data[3 * i + 2] = p[0]; }
        else { img.data[3 * i] = p[2]; img.data[3 * i + 1] = p[1]; img.data[3 * i + 2] = p[0]; }     // RGB -> BGR
    }
    return true;
}

inline bool read_pnm(const std::vector<uint8_t> &buf, Image &img, bool force_color)
{
    if (buf.size() < 3 || buf[0] != 'P' || (buf[1] != '5' && buf[1] != '6')) return false;
    size_t pos = 2;
    int vals[3], nv = 0;
    while (nv < 3 && pos < buf.size()) {
        while (pos < buf.size() && std::isspace(buf[pos])) ++pos;
        if (pos < buf.size() && buf[pos] == '#') { while (pos < buf.size() && buf[pos] != '\n') ++pos; continue; }
        int v = 0; bool any = false;
        while (pos < buf.size() && std::isdigit(buf[pos])) { v = v * 10 + (buf[pos] - '0'); ++pos; any = true; }
        if (!any) return false;
        vals[nv++] = v;
    }
    ++pos;                                                             // single whitespace after maxval
    const int spp = buf[1] == '6' ? 3 : 1;
    if (nv < 3 || vals[2] != 255 || pos + (size_t)vals[0] * vals[1] * spp > buf.size()) return false;
    img.cols = vals[0]; img.rows = vals[1]; img.channels = (spp == 1 && !force_color) ? 1 : 3;
    const size_t n = (size_t)img.rows * img.cols;
    img.data.resize(n * img.channels);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *p = &buf[pos + i * spp];
        if (img.channels == 1) img.data[i] = p[0];
        else if (spp == 1) { img.data[3 * i] = img.data[3 * i + 1] = img.data[3 * i + 2] = p[0]; }
        else { img.data[3 * i] = p[2]; img.data[3 * i + 1] = p[1]; img.data[3 * i + 2] = p[0]; }
    }
    return true;
}

// imread(path, CV_LOAD_IMAGE_COLOR) when force_color, else as stored
inline bool imread(const std::string &path, Image &img, bool force_color = true)
{
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) return false;
    if (read_png(buf, img, force_color) || read_pnm(buf, img, force_color)) return true;
    return jpeg::decode(buf.data(), buf.size(), img.rows, img.cols, img.channels, img.data, force_color);
}

inline void put_chunk(std::vector<uint8_t> &out, const char *type, const uint8_t *d, uint32_t len)
{
    const uint8_t l[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    out.insert(out.end(), l, l + 4);
    const size_t s = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), d, d + len);
    const uint32_t c = (uint32_t)crc32(0L, &out[s], (uInt)(len + 4));
    const uint8_t cb[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
    out.insert(out.end(), cb, cb + 4);
}

inline bool imwrite(const std::string &path, const Image &img)
{
    const size_t n = (size_t)img.rows * img.cols;
    std::vector<uint8_t> out;
    if (ends_with(path, ".jpg") || ends_with(path, ".jpeg")) {
        if (!jpeg::encode(img.data.data(), img.rows, img.cols, img.channels, 95, out)) return false;     // cv::imwrite's default quality
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) return false;
        const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
        std::fclose(f);
        return ok;
    }
    if (ends_with(path, ".png")) {
        const int spp = img.channels;
        std::vector<uint8_t> raw(((size_t)img.cols * spp + 1) * img.rows);
        for (int y = 0; y < img.rows; ++y) {
            uint8_t *r = &raw[(size_t)y * ((size_t)img.cols * spp + 1)];
            r[0] = 0;
            for (int x = 0; x < img.cols; ++x) {
                const uint8_t *p = &img.data[((size_t)y * img.cols + x) * spp];
                if (spp == 1) r[1 + x] = p[0];
                else { r[1 + 3 * x] = p[2]; r[2 + 3 * x] = p[1]; r[3 + 3 * x] = p[0]; }     // BGR -> RGB
            }
        }
        uLongf zl = compressBound((uLong)raw.size());
        std::vector<uint8_t> z(zl);
        if (compress2(z.data(), &zl, raw.data(), (uLong)raw.size(), 3) != Z_OK) return false;
        static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
        out.insert(out.end(), sig, sig + 8);
        uint8_t ih[13] = {(uint8_t)(img.cols >> 24), (uint8_t)(img.cols >> 16), (uint8_t)(img.cols >> 8), (uint8_t)img.cols,
                          (uint8_t)(img.rows >> 24), (uint8_t)(img.rows >> 16), (uint8_t)(img.rows >> 8), (uint8_t)img.rows,
                          8, (uint8_t)(spp == 1 ? 0 : 2), 0, 0, 0};
        put_chunk(out, "IHDR", ih, 13);
        put_chunk(out, "IDAT", z.data(), (uint32_t)zl);
        put_chunk(out, "IEND", nullptr, 0);
    } else {
        char hdr[64];
        const int hl = std::snprintf(hdr, sizeof hdr, "P%c\n%d %d\n255\n", img.channels == 1 ? '5' : '6', img.cols, img.rows);
        out.insert(out.end(), hdr, hdr + hl);
        for (size_t i = 0; i < n; ++i) {
            const uint8_t *p = &img.data[i * img.channels];
            if (img.channels == 1) out.push_back(p[0]);
            else { out.push_back(p[2]); out.push_back(p[1]); out.push_back(p[0]); }
        }
    }
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

}  // namespace imgio
