// Motion-JPEG in an AVI (RIFF) container: the one video format the CLIs can read without a codec library
// (the reference opens any container through cv::VideoCapture, modules/videostrip/src/main.cpp:231-245).
// Frames are the '##dc' / '##db' chunks of the 'movi' list, each a baseline JPEG (decoded by cli/jpeg.hpp; MJPEG
// streams may omit the Huffman tables, for which the decoder falls back to the standard ones).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "imgio.hpp"

namespace avi {

struct Reader {
    std::vector<uint8_t> buf;
    std::vector<std::pair<size_t, size_t>> frames;     // (offset, length) of every video chunk
    double fps = 0.0;
    int width = 0, height = 0;

    static uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

    bool open(const std::string &path)
    {
        if (!imgio::read_file(path, buf) || buf.size() < 12) return false;
        if (std::memcmp(&buf[0], "RIFF", 4) != 0 || std::memcmp(&buf[8], "AVI ", 4) != 0) return false;
        walk(12, buf.size(), 0);
        return !frames.empty();
    }
    // LIST chunks nest at most three deep in a real AVI (RIFF > movi > rec); a crafted file may not recurse further
    static constexpr int MAX_DEPTH = 8;
    void walk(size_t pos, size_t end, int depth)
    {
        while (pos + 8 <= end) {
            const uint32_t len = le32(&buf[pos + 4]);
            const size_t body = pos + 8, next = body + len + (len & 1);
            if (body + len > end) break;               // a child may not overrun its parent
            if (std::memcmp(&buf[pos], "LIST", 4) == 0 && len >= 4) {
                if (depth < MAX_DEPTH) walk(body + 4, body + len, depth + 1);
            } else if (std::memcmp(&buf[pos], "avih", 4) == 0 && len >= 40) {
                const uint32_t us = le32(&buf[body]);
                if (us) fps = 1e6 / (double)us;
                width = (int)le32(&buf[body + 32]); height = (int)le32(&buf[body + 36]);
            } else if (len > 0 && buf[pos + 2] == 'd' && (buf[pos + 3] == 'c' || buf[pos + 3] == 'b') &&
                       buf[pos] >= '0' && buf[pos] <= '9' && buf[pos + 1] >= '0' && buf[pos + 1] <= '9') {
                frames.emplace_back(body, (size_t)len);
            }
            pos = next;
        }
    }
    size_t count() const { return frames.size(); }
    bool read(size_t i, imgio::Image &img) const
    {
        if (i >= frames.size()) return false;
        return jpeg::decode(&buf[frames[i].first], frames[i].second, img.rows, img.cols, img.channels, img.data, true);
    }
};

}  // namespace avi
