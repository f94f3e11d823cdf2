// refshim_check -- exercises the reference-signature shims of include/uwip.hpp (uw::ref::*) the way a call site of the
// reference would: globals videoWidth / videoHeight, keyframe*, by-value Mats, default percentiles.
//   refshim_check <key.png> <object.png>   prints:  stretch <sum of bytes>  blur <v>  overlap <v>  area <v>
#include "cliutil.hpp"
using namespace uw::ref;     // the reference's names, unqualified

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    imgio::Image a, b;
    if (!imgio::imread(argv[1], a, true) || !imgio::imread(argv[2], b, true)) return 3;
    try {
        videoWidth = a.cols; videoHeight = a.rows;                 // main.cpp:238-239
        uw::keyframe kf;
        kf.img = as_mat(a);
        kf.new_img = true;
        const float ov = calcOverlap(&kf, as_mat(b));              // videostrip.hpp:84
        const float bl = calcBlur(as_mat(b));                      // videostrip.hpp:98 (frame already 640 wide here)
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const float ar = overlapArea(I3);                          // videostrip.hpp:118
        uw::Mat m = as_mat(b);
        imgChannelStretch(m, m, 2, 98);                            // preprocessing.h:66 (lane 0 of the packed image)
        unsigned long long s = 0;
        for (size_t i = 0; i < b.data.size(); ++i) s += b.data[i];
        float hist[256];
        getHistogram(&m, hist);                                    // preprocessing.h:38
        std::printf("stretch %llu hist0 %.0f blur %.6f overlap %.6f area %.6f ch %d %d\n", s, hist[0], bl, ov, ar, numChannel('G'), numSpace('V'));
        uw::Videostrip::releaseKeyframe(kf);
    } catch (const uw::Error &e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
