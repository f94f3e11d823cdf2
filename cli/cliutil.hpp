// Tiny helpers shared by the CLIs: `-key=value` parsing in the style of cv::CommandLineParser
// (modules/histretch/src/histretch.cpp:68-77) and wall-clock timing (getTickCount, :165,257-261).
#pragma once
#include <chrono>
#include <map>
#include <string>
#include <vector>
#include "imgio.hpp"
#include "uwip.hpp"

struct Args {
    std::map<std::string, std::string> kv;     // -c=RGB  -> kv["c"] = "RGB";  -r -> kv["r"] = ""
    std::vector<std::string> pos;
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    std::string get(const std::string &k, const std::string &def) const { auto it = kv.find(k); return it == kv.end() ? def : it->second; }
};

// `spaced` lists flags that take their value as the NEXT argument (args.hxx style: -k 11)
inline Args parse_args(int argc, char **argv, const std::vector<std::string> &spaced = {})
{
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        if (s.size() > 1 && s[0] == '-' && !(s.size() > 1 && std::isdigit((unsigned char)s[1]))) {
            size_t b = s.find_first_not_of('-');
            std::string body = s.substr(b);
            size_t eq = body.find('=');
            std::string key = eq == std::string::npos ? body : body.substr(0, eq);
            std::string val = eq == std::string::npos ? "" : body.substr(eq + 1);
            bool takes_next = false;
            for (auto &k : spaced) takes_next = takes_next || k == key;
            if (eq == std::string::npos && takes_next && i + 1 < argc) val = argv[++i];
            a.kv[key] = val;
        } else {
            a.pos.push_back(s);
        }
    }
    return a;
}

inline uw::Mat as_mat(imgio::Image &im)
{
    uw::Mat m;
    m.data = im.data.data(); m.rows = im.rows; m.cols = im.cols; m.chans = im.channels; m.step = (size_t)im.cols * im.channels;
    return m;
}

struct Stopwatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
