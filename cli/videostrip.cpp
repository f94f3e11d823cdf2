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
#include "cliutil.hpp"
#include "avi.hpp"

static const int TARGET_WIDTH = 640;          // videostrip.hpp:48
static const double OVERLAP_MIN = 0.4;        // videostrip.hpp:50
static const int DEFAULT_KWINDOW = 11;        // videostrip.hpp:51

int main(int argc, char **argv)
{
    const Args a = parse_args(argc, argv, {"k", "windowSize", "s", "timeSkip", "p", "minOverlap"});
    if (a.pos.size() < 2 || a.has("h") || a.has("help")) {
        std::printf("videostrip - smart extraction of video frames\n"
                    "usage: videostrip [-k windowSize] [-s timeSkip] [-p minOverlap] [-r] [--png] <video.avi (Motion-JPEG) | frame_list.txt> <output_prefix>\n");
        return 0;
    }
    const int kWindow = std::atoi(a.get("k", a.get("windowSize", std::to_string(DEFAULT_KWINDOW))).c_str());
    const int skip = std::atoi(a.get("s", a.get("timeSkip", "0")).c_str());
    const double minOverlap = std::atof(a.get("p", a.get("minOverlap", std::to_string(OVERLAP_MIN))).c_str());
    const std::string InputFile = a.pos[0], OutputFile = a.pos[1];
    const char *ext = a.has("png") ? "png" : "jpg";
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
    auto read_frame = [&](imgio::Image &im) {
        if (next >= frames.size()) return false;
        const size_t i = next++;
        return is_avi ? video.read(i, im) : imgio::imread(frames[i], im, true);
    };

    try {
        uw::Context ctx(0);
        uw::Videostrip vsx(ctx);
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
