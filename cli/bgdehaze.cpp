// bgdehaze -- Underwater Image Restoration by Blue-Green Channels Dehazing and Red Channel Correction.
// The reference is `python main.py -i <index> -w <window>` over a file list from a missing util.py
// (modules/bgdehaze/main.py:11,22-33); this takes the files directly:
//   bgdehaze [-w N] [--rc] [--guard-s] [--histretch LETTERS] <input> <output>
//   -w N          window size of the dark channel (default 15, main.py:28-29)
//   --rc          stop after RC_correction (BGDehaze.py:59-69) instead of adaptiveExp_map (:71-89)
//   --guard-s     S = 1 where BGDehaze.py:83 divides 0 by 0 (a deviation; the default is the reference's behaviour: the
//                 NaN spreads over the frame, which comes out black -- SURVEY B-11; --as-written is accepted and names it)
//   --histretch L chain `histretch -c=L` (2/98 percent) on the result in the same run, e.g. --histretch RGB
#include "cliutil.hpp"

int main(int argc, char **argv)
{
    const Args a = parse_args(argc, argv, {"w", "window", "histretch"});
    if (a.pos.size() < 2 || a.has("help") || a.has("h")) {
        std::printf("usage: bgdehaze [-w N] [--rc] [--guard-s] [--histretch LETTERS] <input> <output>\n");
        return 0;
    }
    const int w = std::atoi(a.get("w", a.get("window", "15")).c_str());
    std::printf("processing %s...\n", a.pos[0].c_str());                                  // main.py:15
    imgio::Image src;
    if (!imgio::imread(a.pos[0], src, true)) { std::printf("cannot read %s\n", a.pos[0].c_str()); return -1; }
    try {
        uw::Context ctx(0);
        uw::DeviceMat in(ctx, as_mat(src));
        uw::DeviceMat out(ctx, src.rows, src.cols, 3);
        const int flags = (a.has("rc") ? 0 : UWIP_DEHAZE_FULL) | (a.has("guard-s") ? UWIP_DEHAZE_GUARD_S : 0);
        if (a.has("histretch"))
            ctx.check(uwip_dehaze_histretch(ctx.get(), in.batch(), out.batch(), w, flags, a.get("histretch", "RGB").c_str(), 2, 98, 0u));
        else
            ctx.check(uwip_dehaze(ctx.get(), in.batch(), out.batch(), w, flags, nullptr, nullptr, nullptr));
        uw::Mat m = as_mat(src);
        out.download(m);
    } catch (const uw::Error &e) {
        std::printf("error: %s\n", e.what());
        return -1;
    }
    if (!imgio::imwrite(a.pos[1], src)) { std::printf("cannot write %s\n", a.pos[1].c_str()); return -1; }
    std::printf("saved %s\n", a.pos[1].c_str());                                          // main.py:20
    return 0;
}
