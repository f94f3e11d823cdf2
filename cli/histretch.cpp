// histretch -- percentile histogram stretch of selected channels.
// Flags and messages follow modules/histretch/src/histretch.cpp:61-272:
//   histretch [-c=<letters>] [-cuda=0|1] [-time=0|1] [--fixed-order] <input> <output>
// Differences, on purpose: headless (no imshow / waitKey), -time is optional
// (B-4), and there is no CPU implementation in this build: -cuda=0 is refused (BASELINE config 1, the
// reference's OpenCV CPU path on one 640x480 PNG, is therefore exercised through the HIP path: tests/test_cli.py).
// --fixed-order: letters of HSV / hsl / Lab / YCX keep their stretch (merge before converting back) instead of the
// as-written round trip (B-3).
#include "cliutil.hpp"

int main(int argc, char **argv)
{
    const Args a = parse_args(argc, argv);
    std::printf("histretch (uwip-mi355x) -- %s\n", uwip_version());
    if (argc < 3 || a.has("help") || a.has("h") || a.pos.size() < 2) {
        std::printf("C++ implementation of Histogram Stretching for specific channels of input image\n"
                    "usage: histretch [-c=<channels>] [-cuda=0|1] [-time=0|1] <input> <output>\n"
                    "Argument 'c=<channels>' is a string containing an ordered list of desired channels to be stretched\n"
                    "\t-c=R|G|B\tfor RGB space\n\t-c=H|S|V\tfor HSV space\n\t-c=h|s|l\tfor hsl space\n\t-c=L|a|b\tfor Lab space\n"
                    "\t-c=Y|C|X\tfor YCrCb space\n\t--fixed-order\tkeep the stretch of the non-RGB letters (the reference discards it)\n"
                    "\t--opencv32\tLab -> BGR as OpenCV 3.2 does (float form) instead of OpenCV 3.4.x (integer form, the default)\n"
                    "\tExample:\n\t$ histretch -c=RGB input.png output.png -cuda=1 -time=1\n");
        return 0;
    }
    const std::string cChannel = a.get("c", "r");                 // default "r" is a no-op, as in the reference (B-1)
    const int Time = std::atoi(a.get("time", "0").c_str());
    const int CUDA = std::atoi(a.get("cuda", "1").c_str());
    std::printf("***************************************\nInput: %s\nOutput: %s\nChannel: %s\n", a.pos[0].c_str(), a.pos[1].c_str(), cChannel.c_str());
    if (CUDA == 0) {
        std::printf("CUDA deactivated\nExiting... this build has no CPU implementation (use the reference for -cuda=0)\n");
        return -1;
    }
    if (uw::Context::deviceCount() <= 0) { std::printf("No HIP device detected\n"); return -1; }
    imgio::Image src;
    if (!imgio::imread(a.pos[0], src, true)) { std::printf("Failed to read input image, exiting...\n"); return -1; }
    std::printf("Applying %zu histretch\n", cChannel.size());
    try {
        uw::Context ctx(0);
        for (size_t nc = 0; nc < cChannel.size(); ++nc) {
            std::printf("\tChannel[%zu]: %c\n", nc, cChannel[nc]);
            if (uw::numSpace(cChannel[nc]) == -1) std::printf("Option %c not recognized, skipping...\n", cChannel[nc]);
        }
        Stopwatch sw;
        uw::histretch(ctx, as_mat(src), cChannel, 2, 98, a.has("fixed-order"), a.has("opencv32"));   // min_percent = 2, max_percent = 98 (histretch.cpp:154)
        if (Time == 1) std::printf("\nExecution Time GPU :%g ms \n", sw.ms());
    } catch (const uw::Error &e) {
        std::printf("error: %s\n", e.what());
        return -1;
    }
    std::printf("hS: saving to disk\n");
    if (!imgio::imwrite(a.pos[1], src)) { std::printf("Failed to write %s\n", a.pos[1].c_str()); return -1; }
    return 0;
}
