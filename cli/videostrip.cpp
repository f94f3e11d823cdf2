// videostrip -- smart extraction of frames by estimated overlap.
// Flags follow modules/videostrip/include/options.h:16-24 / src/main.cpp:55-399:
//   videostrip [-k N] [-s N] [-p OVERLAP] [-r] <input> <output_prefix>
// <input> is a Motion-JPEG .avi (the one container read without a codec library: cli/avi.hpp; -s skips that many
// SECONDS of it, main.cpp:247-255) or a text file listing one frame image per line (PNG / PPM / JPEG; -s then skips
// that many FRAMES).  Output: <prefix>NNNN.jpg as the reference writes them (main.cpp:294,377; --png for lossless
// PNG instead) and <prefix>videostrip_report.txt with the reference's TSV columns
// (main.cpp:263,297,381).  The selector loop is main.cpp:300-394 as written, except that the end
// of the input ends the program with exit code 0 (the reference calls exit(EXIT_FAILURE), B-14).
#include <cmath>
#include <fstream>
#include <thread>
#include "cliutil.hpp"
#include "avi.hpp"

static const int TARGET_WIDTH = 640;          // videostrip.hpp:48
static const double OVERLAP_MIN = 0.4;        // videostrip.hpp:50
static const int DEFAULT_KWINDOW = 11;        // videostrip.hpp:51

// ---- --gpus N: the selector across the GPUs of a node (SURVEY.md 8e option 1; main.cpp:216 leaves multi-device as a TODO) ----
// What is sequential in main.cpp:300-394 is the decision chain; decoding, resize, detect, describe and the blur metric
// depend on one frame only.  N worker threads (one context per GPU, frame slices contiguous) extract per frame the cached
// part of `struct keyframe` (videostrip.hpp:62-68) and calcBlur; the main thread then replays the chain on GPU 0 with the
// overlaps of the next LOOKAHEAD frames against the current key frame in one matcher launch.  Same rows as the
// frame-by-frame loop below (tests/test_cli.py).
struct FrameRec {
    std::vector<uwip_keypoint> kps;
    std::vector<uint8_t> desc;
    float blur = 0.f;
    bool ok = false;
};

int main(int argc, char **argv)
{
    const Args a = parse_args(argc, argv, {"k", "windowSize", "s", "timeSkip", "p", "minOverlap", "g", "gpus"});
    if (a.pos.size() < 2 || a.has("h") || a.has("help")) {
        std::printf("videostrip - smart extraction of video frames\n"
                    "usage: videostrip [-k windowSize] [-s timeSkip] [-p minOverlap] [-r] [--png] [--min6] [--relative-threshold] [-g gpus] <video.avi (Motion-JPEG) | frame_list.txt> <output_prefix>\n"
                    "  default: any homography found from >= 4 good matches counts, as in the reference (videostrip.cpp:252-272)\n"
                    "  --min6  a homography needs >= 6 RANSAC inliers, else -2.0 (a deviation; --min4 is accepted and names the default)\n"
                    "  default: fixed detector threshold, as SURF's hessianThreshold is fixed (videostrip.cpp:206)\n"
                    "  --relative-threshold  detector threshold relative to the frame's contrast: raw frames of turbid water have no\n"
                    "          response above a fixed threshold (a deviation; use it together with --min6)\n");
        return 0;
    }
    const int kWindow = std::atoi(a.get("k", a.get("windowSize", std::to_string(DEFAULT_KWINDOW))).c_str());
    const int skip = std::atoi(a.get("s", a.get("timeSkip", "0")).c_str());
    const double minOverlap = std::atof(a.get("p", a.get("minOverlap", std::to_string(OVERLAP_MIN))).c_str());
    const std::string InputFile = a.pos[0], OutputFile = a.pos[1];
    const char *ext = a.has("png") ? "png" : "jpg";
    const unsigned match_flags = a.has("min6") ? UWIP_OVERLAP_MIN6 : 0u;
    const unsigned detect_flags = a.has("relative-threshold") ? UWIP_OVERLAP_RELATIVE_THRESHOLD : 0u;
    std::vector<std::string> frames;
    avi::Reader video;
    const bool is_avi = imgio::ends_with(InputFile, ".avi");
    if (is_avi) {
        if (!video.open(InputFile)) { std::printf("Unable to open: %s\n", InputFile.c_str()); return EXIT_FAILURE; }   // main.cpp:232-235
        frames.resize(video.count());
    } else {
        std::ifstream f(InputFile); std::string l; while (std::getline(f, l)) if (!l.empty()) frames.push_back(l);
    }
    if (frames.empty()) { std::printf("Unable to open frame list: %s\n", InputFile.c_str()); return EXIT_FAILURE; }
    std::ofstream report(OutputFile + "videostrip_report.txt");                          // main.cpp:120-127
    report << "Input:\t" << InputFile << "\n";
    // timeSkip is in seconds of video (main.cpp:247-255: frames = fps * seconds); a frame list has no rate: frames
    size_t next = (size_t)std::max(is_avi && video.fps > 0 ? (int)(video.fps * skip) : skip, 0);
    auto read_at = [&](size_t i, imgio::Image &im) { return is_avi ? video.read(i, im) : imgio::imread(frames[i], im, true); };
    auto read_frame = [&](imgio::Image &im) {
        if (next >= frames.size()) return false;
        return read_at(next++, im);
    };
    const int gpus = std::atoi(a.get("g", a.get("gpus", "0")).c_str());
    if (gpus >= 1) {
        try {
            const size_t first = next, n = frames.size() > first ? frames.size() - first : 0;
            if (n == 0) { std::printf("Unable to read first frame\n"); return EXIT_FAILURE; }
            const int ndev = std::max(1, uw::Context::deviceCount());
            std::vector<FrameRec> rec(n);
            int vw = 0, vh = 0;
            {
                imgio::Image f0;
                if (!read_at(first, f0)) { std::printf("Unable to read first frame\n"); return EXIT_FAILURE; }
                vw = f0.cols; vh = f0.rows;
            }
            std::vector<std::thread> th;
            std::vector<std::string> errs(gpus);
            for (int t = 0; t < gpus; ++t)
                th.emplace_back([&, t] {
                    try {
                        uw::Context ctx(t % ndev);
                        uw::Videostrip vsx(ctx);
                        uwip_features *f = nullptr;
                        ctx.check(uwip_features_create(ctx.get(), 1, &f));
                        const size_t a0 = n * t / gpus, a1 = n * (t + 1) / gpus;          // contiguous slices
                        imgio::Image im;
                        uw::Mat res;
                        std::vector<uint8_t> store;
                        std::vector<uwip_keypoint> kps(UWIP_MAX_KEYPOINTS);
                        std::vector<uint8_t> desc((size_t)UWIP_MAX_KEYPOINTS * 64);
                        for (size_t i = a0; i < a1; ++i) {
                            if (!read_at(first + i, im) || im.cols != vw || im.rows != vh) continue;
                            uw::DeviceMat d(ctx, as_mat(im));
                            ctx.check(uwip_overlap_detect_ex(ctx.get(), d.batch(), f, 0, detect_flags));
                            int32_t cnt = 0;
                            ctx.check(uwip_features_download(ctx.get(), f, 0, kps.data(), desc.data(), &cnt));
                            rec[i].kps.assign(kps.begin(), kps.begin() + cnt);
                            rec[i].desc.assign(desc.begin(), desc.begin() + (size_t)cnt * 64);
                            vsx.resize(as_mat(im), res, store);
                            rec[i].blur = vsx.calcBlur(res);
                            rec[i].ok = true;
                        }
                        uwip_features_destroy(f);
                    } catch (const uw::Error &e) { errs[t] = e.what(); }
                });
            for (auto &x : th) x.join();
            for (auto &e : errs) if (!e.empty()) { std::printf("error: %s\n", e.c_str()); return EXIT_FAILURE; }
            size_t nn = 0;
            while (nn < n && rec[nn].ok) ++nn;                 // an unreadable frame ends the input, as in the loop below
            if (nn == 0) { std::printf("Unable to read first frame\n"); return EXIT_FAILURE; }
            // ---- the decision chain on GPU 0
            const int LOOKAHEAD = 8;
            uw::Context ctx(0);
            uwip_features *fs = nullptr;
            ctx.check(uwip_features_create(ctx.get(), 1 + LOOKAHEAD, &fs));
            float *d_ratio = nullptr;
            ctx.check(uwip_malloc(ctx.get(), sizeof(float) * LOOKAHEAD, (void **)&d_ratio));
            int oh = 0, ow = 0;
            ctx.check(uwip_overlap_working_size(vh, vw, &oh, &ow));
            const uwip_keypoint no_kp{};
            const uint8_t no_desc[64] = {0};
            auto upload = [&](int slot, const FrameRec &r) {          // a frame without keypoints uploads an empty slot
                ctx.check(uwip_features_upload(ctx.get(), fs, slot, oh, ow, r.kps.empty() ? &no_kp : r.kps.data(),
                                               r.desc.empty() ? no_desc : r.desc.data(), (int32_t)r.kps.size()));
            };
            const float hResizeFactor = (float)TARGET_WIDTH / (float)vw;
            std::printf("Video metadata:\n\tSize:\t%d x %d\n\tFrames:\t%zu\n\thResize:\t%g\nTarget minOverlap:\t%g\nWindow size:\t%d\n",
                        vw, vh, frames.size(), hResizeFactor, minOverlap, kWindow);
            report << "Video metadata:\n\tSize:\t" << vw << " x " << vh << "\n\tFrames:\t" << frames.size()
                   << "\n\thResize:\t" << hResizeFactor << "\nTarget minOverlap:\t" << minOverlap << "\nWindow size:\t" << kWindow << "\n";
            report << "***************************************\nID\tFrame\tFilename\tOverlap\tBlur\n";
            char name[512];
            imgio::Image out_img;
            int out_frame = 0;
            std::snprintf(name, sizeof name, "%s%04d.%s", OutputFile.c_str(), out_frame, ext);
            read_at(first, out_img);
            imgio::imwrite(name, out_img);
            report << "0\t0\t" << name << "\t0.0\t0.0\n";
            size_t key = 0, nxt = 1;
            int read_frames = 1;
            size_t spec_key = (size_t)-1, spec_at = 0, spec_n = 0;
            float spec[LOOKAHEAD];
            upload(0, rec[0]);
            while (nxt < nn) {
                if (spec_key != key || !(spec_at <= nxt && nxt < spec_at + spec_n)) {
                    const int m = (int)std::min<size_t>(LOOKAHEAD, nn - nxt);
                    int32_t pq[LOOKAHEAD], pt[LOOKAHEAD];
                    for (int j = 0; j < m; ++j) { upload(1 + j, rec[nxt + j]); pq[j] = 1 + j; pt[j] = 0; }
                    ctx.check(uwip_overlap_match_ex(ctx.get(), fs, fs, pq, pt, m, vw, vh, 1, match_flags, d_ratio, nullptr, nullptr, nullptr, nullptr));
                    ctx.check(uwip_memcpy_d2h(ctx.get(), spec, d_ratio, sizeof(float) * m));
                    spec_key = key; spec_at = nxt; spec_n = (size_t)m;
                }
                float currOverlap = spec[nxt - spec_at];
                const size_t cur = nxt;
                nxt++; read_frames++;
                std::printf("\rFrame: %d\tOverlap: %g", read_frames - 1, currOverlap);
                if (currOverlap == -2.0f) currOverlap = (float)(OVERLAP_MIN + 0.01);          // :321-326
                if (currOverlap <= minOverlap) {                                              // :329
                    float bestBlur = rec[cur].blur;
                    int best_frame_number = (int)(first + nxt) - 1;
                    size_t best = cur;
                    bool eof = false;
                    for (int w = 0; w < kWindow; ++w) {                                       // :344-366
                        if (nxt >= nn) { eof = true; break; }
                        const size_t g = nxt;
                        nxt++; read_frames++;
                        if (rec[g].blur > bestBlur) { bestBlur = rec[g].blur; best = g; best_frame_number = read_frames; }
                    }
                    key = best;
                    upload(0, rec[key]);
                    out_frame++;
                    std::snprintf(name, sizeof name, "%s%04d.%s", OutputFile.c_str(), out_frame, ext);
                    read_at(first + key, out_img);
                    imgio::imwrite(name, out_img);
                    std::printf("\nExported frame: %d [%d]\n", best_frame_number, out_frame);
                    report << out_frame << "\t" << best_frame_number << "\t" << name << "\t" << currOverlap << "\t" << bestBlur << "\n";
                    if (eof) break;
                }
            }
            std::printf("\nEnd of input.\n");
            uwip_free(ctx.get(), d_ratio);
            uwip_features_destroy(fs);
        } catch (const uw::Error &e) {
            std::printf("error: %s\n", e.what());
            return EXIT_FAILURE;
        }
        return 0;
    }

    try {
        uw::Context ctx(0);
        uw::Videostrip vsx(ctx);
        vsx.match_flags = match_flags;
        vsx.detect_flags = detect_flags;
        uw::keyframe kframe;
        imgio::Image kimg, frame, bestframe;
        uw::Mat res;                              // res_frame: cv::resize(frame, res_frame, Size(), f, f), main.cpp:311
        std::vector<uint8_t> res_store;
        int out_frame = 0, read_frames = 0;
        if (!read_frame(kimg)) { std::printf("Unable to read first frame\n"); return EXIT_FAILURE; }
        read_frames++;
        vsx.videoWidth = kimg.cols; vsx.videoHeight = kimg.rows;                          // main.cpp:238-239
        const float hResizeFactor = (float)TARGET_WIDTH / (float)kimg.cols;               // :242
        std::printf("Video metadata:\n\tSize:\t%d x %d\n\tFrames:\t%zu\n\thResize:\t%g\nTarget minOverlap:\t%g\nWindow size:\t%d\n",
                    kimg.cols, kimg.rows, frames.size(), hResizeFactor, minOverlap, kWindow);
        report << "Video metadata:\n\tSize:\t" << kimg.cols << " x " << kimg.rows << "\n\tFrames:\t" << frames.size()
               << "\n\thResize:\t" << hResizeFactor << "\nTarget minOverlap:\t" << minOverlap << "\nWindow size:\t" << kWindow << "\n";
        report << "***************************************\nID\tFrame\tFilename\tOverlap\tBlur\n";
        kframe.img = as_mat(kimg);
        kframe.new_img = true;
        char name[512];
        std::snprintf(name, sizeof name, "%s%04d.%s", OutputFile.c_str(), out_frame, ext);   // :293-297
        imgio::imwrite(name, kimg);
        report << "0\t0\t" << name << "\t0.0\t0.0\n";
        for (;;) {                                                                        // :300
            if (!read_frame(frame)) { std::printf("\nEnd of input.\n"); break; }          // :303-307 (B-14)
            read_frames++;
            float currOverlap = vsx.calcOverlap(&kframe, as_mat(frame));                  // :315
            std::printf("\rFrame: %d\tOverlap: %g", read_frames - 1, currOverlap);
            if (currOverlap == -2.0f) currOverlap = (float)(OVERLAP_MIN + 0.01);          // :321-326
            if (currOverlap <= minOverlap) {                                              // :329
                std::printf("\n");
                vsx.resize(as_mat(frame), res, res_store);
                float bestBlur = vsx.calcBlur(res);                                       // :338
                int best_frame_number = (int)next - 1;
                bestframe = frame;
                bool eof = false;
                for (int n = 0; n < kWindow; ++n) {                                       // :344-366
                    if (!read_frame(frame)) { eof = true; break; }
                    read_frames++;
                    vsx.resize(as_mat(frame), res, res_store);
                    const float currBlur = vsx.calcBlur(res);
                    std::printf("\rRefining search [%d/%d]\tBlur: %g\tBest: %g", n + 1, kWindow, currBlur, bestBlur);
                    if (currBlur > bestBlur) { bestBlur = currBlur; bestframe = frame; best_frame_number = read_frames; }
                }
                kimg = bestframe;                                                         // :368-369
                kframe.img = as_mat(kimg);
                kframe.new_img = true;
                out_frame++;
                std::snprintf(name, sizeof name, "%s%04d.%s", OutputFile.c_str(), out_frame, ext);
                imgio::imwrite(name, kimg);
                std::printf("\nExported frame: %d [%d]\n", best_frame_number, out_frame);
                report << out_frame << "\t" << best_frame_number << "\t" << name << "\t" << currOverlap << "\t" << bestBlur << "\n";   // :381
                std::printf("*************\n");
                if (eof) { std::printf("End of input.\n"); break; }
            }
        }
        uw::Videostrip::releaseKeyframe(kframe);
    } catch (const uw::Error &e) {
        std::printf("error: %s\n", e.what());
        return EXIT_FAILURE;
    }
    return 0;
}
