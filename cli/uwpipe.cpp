// uwpipe -- the reference's four tools run back to back as ONE program over the C ABI's whole-chain entry (uwip_pipe_*):
//   bgdehaze (modules/bgdehaze/main.py:14-20) -> histretch -c=RGB (modules/histretch/src/histretch.cpp:217-254) ->
//   aclahe (modules/aclahe/src/aclahe.cpp:152-218 + python/ACLAHE.py:9-129, python/main.py:19-20) ->
//   calcOverlap of every frame against its predecessor (modules/videostrip/src/videostrip.cpp:192-289)
// on the frames of a Motion-JPEG .avi or of a frame list, in batches through page-locked host buffers
// (uwip_pipe_step_host: batch k + 1 is uploaded and batch k - 1 leaves while batch k's kernels run).
//   uwpipe [-b N] [-c LETTERS] [-w N] [--guard-s] [--min6] [--relative-threshold] [--png] <video.avi | frame_list.txt> <output_prefix>
// writes <prefix>NNNN.jpg (the enhanced frames) and <prefix>uwpipe_report.txt (TSV: ID, Filename, Overlap, BS, CL).
// Defaults are the reference's rules (uwip_pipe_config_default); the three switches are the library's opt-in deviations.
#include <algorithm>
#include <cstring>
#include <fstream>
#include "avi.hpp"
#include "cliutil.hpp"

int main(int argc, char **argv)
{
    const Args a = parse_args(argc, argv, {"b", "batch", "c", "w", "window"});
    if (a.pos.size() < 2 || a.has("h") || a.has("help")) {
        std::printf("uwpipe - bgdehaze -> histretch -> aclahe -> overlap of every frame against its predecessor\n"
                    "usage: uwpipe [-b N] [-c LETTERS] [-w N] [--guard-s] [--min6] [--relative-threshold] [--png] <video.avi (Motion-JPEG) | frame_list.txt> <output_prefix>\n"
                    "  -b N      frames per step (default 8)\n"
                    "  -c L      histretch letters (default RGB)\n"
                    "  -w N      bgdehaze window (default 15)\n"
                    "  --guard-s / --min6 / --relative-threshold   the library's opt-in deviations from the reference's rules (uwip.h)\n");
        return 0;
    }
    const std::string InputFile = a.pos[0], OutputFile = a.pos[1];
    const char *ext = a.has("png") ? "png" : "jpg";
    std::vector<std::string> frames;
    avi::Reader video;
    const bool is_avi = imgio::ends_with(InputFile, ".avi");
    if (is_avi) {
        if (!video.open(InputFile)) { std::printf("Unable to open: %s\n", InputFile.c_str()); return EXIT_FAILURE; }
        frames.resize(video.count());
    } else {
        std::ifstream f(InputFile); std::string l; while (std::getline(f, l)) if (!l.empty()) frames.push_back(l);
    }
    if (frames.empty()) { std::printf("Unable to open frame list: %s\n", InputFile.c_str()); return EXIT_FAILURE; }
    auto read_at = [&](size_t i, imgio::Image &im) { return is_avi ? video.read(i, im) : imgio::imread(frames[i], im, true); };
    imgio::Image first;
    if (!read_at(0, first)) { std::printf("Unable to read first frame\n"); return EXIT_FAILURE; }
    const int rows = first.rows, cols = first.cols;
    const size_t n = frames.size();
    const int B = (int)std::min<size_t>(n, (size_t)std::max(1, std::atoi(a.get("b", a.get("batch", "8")).c_str())));
    const size_t fbytes = (size_t)rows * cols * 3;

    uwip_ctx *ctx = nullptr;
    uwip_pipe *pipe = nullptr;
    void *h_in[2] = {nullptr, nullptr}, *h_out = nullptr, *h_ratio = nullptr;
    int rc = 0;
    const char *what = "";
#define CK(expr, msg) do { rc = (expr); if (rc) { what = msg; goto fail; } } while (0)
    {
        CK(uwip_ctx_create(0, nullptr, &ctx), "uwip_ctx_create (no HIP device? there is no CPU fallback)");
        uwip_pipe_config cfg;
        uwip_pipe_config_default(&cfg, B, rows, cols);
        std::snprintf(cfg.letters, sizeof cfg.letters, "%s", a.get("c", "RGB").c_str());
        cfg.w = std::atoi(a.get("w", a.get("window", "15")).c_str());
        if (a.has("guard-s")) cfg.dehaze_flags |= UWIP_DEHAZE_GUARD_S;
        if (a.has("min6")) cfg.match_flags |= UWIP_OVERLAP_MIN6;
        if (a.has("relative-threshold")) cfg.detect_flags |= UWIP_OVERLAP_RELATIVE_THRESHOLD;
        CK(uwip_pipe_create(ctx, &cfg, nullptr, &pipe), "uwip_pipe_create");
        for (int s = 0; s < 2; ++s) CK(uwip_host_alloc(ctx, fbytes * B, &h_in[s]), "uwip_host_alloc");
        CK(uwip_host_alloc(ctx, fbytes * B, &h_out), "uwip_host_alloc");
        CK(uwip_host_alloc(ctx, sizeof(float) * B, &h_ratio), "uwip_host_alloc");

        std::ofstream report(OutputFile + "uwpipe_report.txt");
        report << "Input:\t" << InputFile << "\nSize:\t" << cols << " x " << rows << "\nFrames:\t" << n << "\nBatch:\t" << B
               << "\n***************************************\nID\tFilename\tOverlap\tBS\tCL\n";
        // a batch = B consecutive frames; the last one is padded by repeating the final frame (its outputs are dropped)
        auto fill = [&](size_t k, void *dst) -> bool {
            imgio::Image im;
            for (int j = 0; j < B; ++j) {
                const size_t i = std::min(k * B + j, n - 1);
                if (!read_at(i, im) || im.rows != rows || im.cols != cols || im.channels != 3) {
                    std::printf("cannot read frame %zu (or its size differs from the first frame's)\n", i);
                    return false;
                }
                std::memcpy((uint8_t *)dst + fbytes * j, im.data.data(), fbytes);
            }
            return true;
        };
        const size_t nb = (n + B - 1) / B;
        std::vector<int32_t> bs(B), cl(B);
        uint64_t prev_up = 0;                    // upload ticket of the step that last read h_in[(k + 1) & 1]
        if (!fill(0, h_in[0])) { rc = UWIP_ERR_INVALID; what = "reading the input"; goto fail; }
        for (size_t k = 0; k < nb; ++k) {
            const bool more = k + 1 < nb;
            if (more) {
                CK(uwip_pipe_wait(pipe, prev_up), "uwip_pipe_wait");           // h_in[(k + 1) & 1] has left for the device
                if (!fill(k + 1, h_in[(k + 1) & 1])) { rc = UWIP_ERR_INVALID; what = "reading the input"; goto fail; }
            }
            uint64_t t[3];
            CK(uwip_pipe_step_host(pipe, h_in[k & 1], h_out, (float *)h_ratio, more ? h_in[(k + 1) & 1] : nullptr, t), "uwip_pipe_step_host");
            prev_up = t[0];
            CK(uwip_pipe_last_params(pipe, bs.data(), cl.data()), "uwip_pipe_last_params");
            CK(uwip_pipe_wait(pipe, t[1]), "uwip_pipe_wait");
            CK(uwip_pipe_wait(pipe, t[2]), "uwip_pipe_wait");
            for (int j = 0; j < B && k * B + j < n; ++j) {
                const size_t i = k * B + j;
                char name[512];
                std::snprintf(name, sizeof name, "%s%04zu.%s", OutputFile.c_str(), i, ext);
                imgio::Image out;
                out.rows = rows; out.cols = cols; out.channels = 3;
                out.data.assign((uint8_t *)h_out + fbytes * j, (uint8_t *)h_out + fbytes * (j + 1));
                if (!imgio::imwrite(name, out)) { std::printf("cannot write %s\n", name); rc = UWIP_ERR_INVALID; what = "writing"; goto fail; }
                // frame 0 is its own key frame (main.cpp:284-297): its row carries the self-overlap
                report << i << "\t" << name << "\t" << ((float *)h_ratio)[j] << "\t" << bs[j] << "\t" << cl[j] << "\n";
            }
            std::printf("\rbatch %zu / %zu", k + 1, nb);
            std::fflush(stdout);
        }
        std::printf("\nEnd of input.\n");
    }
fail:
    if (rc) std::printf("error: %s: %s\n", what, pipe ? uwip_pipe_last_error(pipe) : (ctx ? uwip_last_error(ctx) : "no context"));
    uwip_pipe_destroy(pipe);
    if (ctx) {
        for (int s = 0; s < 2; ++s) uwip_host_free(ctx, h_in[s]);
        uwip_host_free(ctx, h_out);
        uwip_host_free(ctx, h_ratio);
        uwip_ctx_destroy(ctx);
    }
    return rc ? EXIT_FAILURE : 0;
}
