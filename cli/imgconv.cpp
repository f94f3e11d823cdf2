// imgconv <in> <out> -- decode an image (PNG / PPM / JPEG) and write it in the format of <out>'s extension.
// Used by the tests to check the JPEG codec (cli/jpeg.hpp) against Pillow's libjpeg on the reference's photographs.
#include "imgio.hpp"
int main(int argc, char **argv)
{
    if (argc < 3) { std::printf("usage: imgconv <in> <out> [grey]\n"); return 2; }
    imgio::Image im;
    if (!imgio::imread(argv[1], im, argc < 4)) { std::printf("cannot read %s\n", argv[1]); return 1; }
    if (!imgio::imwrite(argv[2], im)) { std::printf("cannot write %s\n", argv[2]); return 1; }
    std::printf("%d x %d x %d\n", im.cols, im.rows, im.channels);
    return 0;
}
