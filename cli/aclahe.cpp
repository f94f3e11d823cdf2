// aclahe -- entropy-driven automatic CLAHE.   usage: aclahe <input> <output>
// Follows modules/aclahe/src/aclahe.cpp:64-221 and finishes its comment stubs (:209-218) with the
// parameter choice of modules/aclahe/python/ACLAHE.py:66-129: sweep 5 block sizes x 51 clip limits on the
// V plane, print the entropy table (a clean 5 x 51 table, SURVEY B-6), pick (BS, CL), apply CLAHE,
// transform back to BGR, save.
#include "cliutil.hpp"

int main(int argc, char **argv)
{
    const Args a = parse_args(argc, argv);
    std::printf("aclahe (uwip-mi355x) -- %s\n", uwip_version());
    if (a.pos.size() < 2 || a.has("help") || a.has("h")) {
        std::printf("Automatic CLAHE parameter estimation\nusage: aclahe <input> <output>\n");
        return 0;
    }
    std::printf("***************************************\nInput: %s\nOutput: %s\n", a.pos[0].c_str(), a.pos[1].c_str());
    imgio::Image src;
    if (!imgio::imread(a.pos[0], src, true)) { std::printf("Failed to read input image, exiting...\n"); return -1; }
    std::printf("Input image loaded...\n");
    try {
        uw::Context ctx(0);
        uw::DeviceMat bgr(ctx, as_mat(src));
        uw::DeviceMat v(ctx, src.rows, src.cols, 1), out(ctx, src.rows, src.cols, 1);
        ctx.check(uwip_bgr_to_v(ctx.get(), bgr.batch(), v.batch()));                     // aclahe.cpp:152-154
        void *d_tab = nullptr;
        ctx.check(uwip_malloc(ctx.get(), sizeof(float) * 255, &d_tab));
        ctx.check(uwip_aclahe_sweep(ctx.get(), v.batch(), 0, (float *)d_tab));           // :160-193
        float tab[255];
        ctx.check(uwip_memcpy_d2h(ctx.get(), tab, d_tab, sizeof tab));
        uwip_free(ctx.get(), d_tab);
        for (int i = 0; i < 5; ++i) {                                                     // :199-206
            for (int j = 0; j < 51; ++j) std::printf("%g ", tab[i * 51 + j]);
            std::printf("\n");
        }
        int32_t BS = 0, CL = 0;
        ctx.check(uwip_aclahe_auto(ctx.get(), v.batch(), out.batch(), 0, &BS, &CL));      // :209-215
        std::printf("Block size: %d\nClip limit: %d\n", BS, CL);
        ctx.check(uwip_hsv_replace_v(ctx.get(), bgr.batch(), out.batch(), bgr.batch()));  // :216
        uw::Mat m = as_mat(src);
        bgr.download(m);
    } catch (const uw::Error &e) {
        std::printf("error: %s\n", e.what());
        return -1;
    }
    if (!imgio::imwrite(a.pos[1], src)) { std::printf("Failed to write %s\n", a.pos[1].c_str()); return -1; }   // :218
    return 0;
}
