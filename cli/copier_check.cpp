// copier_check -- the host-buffer front end from C++ (uw::Copier over uwip_copier_*): a stream of batches is uploaded from
// page-locked host memory one batch ahead, stretched on the device (uwip_histretch, -c=RGB) and downloaded, every hand-over
// a ticket; the result must equal the synchronous upload -> histretch -> download of the same batches.
//   copier_check <frames per batch> <rows> <cols> <batches>   prints "copier ok <checksum>"
#include <cstdlib>
#include <vector>
#include "uwip.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int F = std::atoi(argv[1]), H = std::atoi(argv[2]), W = std::atoi(argv[3]), NB = std::atoi(argv[4]);
    const size_t bytes = (size_t)F * H * W * 3;
    try {
        uw::Context ctx(0);
        uw::Copier cp(0);
        std::vector<void *> h_in(NB), h_out(NB);
        std::vector<std::vector<uint8_t>> expect(NB);
        uint32_t lcg = 12345u;
        for (int b = 0; b < NB; ++b) {
            ctx.check(uwip_host_alloc(ctx.get(), bytes, &h_in[b]));
            ctx.check(uwip_host_alloc(ctx.get(), bytes, &h_out[b]));
            uint8_t *p = (uint8_t *)h_in[b];
            for (size_t i = 0; i < bytes; ++i) { lcg = lcg * 1664525u + 1013904223u; p[i] = (uint8_t)(40 + ((lcg >> 24) % 150)); }
        }
        void *d[2];
        ctx.check(uwip_malloc(ctx.get(), bytes, &d[0]));
        ctx.check(uwip_malloc(ctx.get(), bytes, &d[1]));
        uwip_batch_u8 bt{};
        bt.rows = H; bt.cols = W; bt.channels = 3; bt.frames = F; bt.step = (size_t)W * 3; bt.frame_stride = bt.step * H;
        // reference result: synchronous copies on the context's own stream
        for (int b = 0; b < NB; ++b) {
            bt.data = d[0];
            ctx.check(uwip_memcpy_h2d(ctx.get(), d[0], h_in[b], bytes));
            ctx.check(uwip_histretch(ctx.get(), &bt, "RGB", 2, 98));
            expect[b].resize(bytes);
            ctx.check(uwip_memcpy_d2h(ctx.get(), expect[b].data(), d[0], bytes));
        }
        // pipelined: batch b + 1 is uploaded while batch b is processed; a device buffer is reused two batches later
        std::vector<uint64_t> t_up(NB, 0), t_dn(NB, 0);
        t_up[0] = cp.upload(d[0], h_in[0], bytes);
        for (int b = 0; b < NB; ++b) {
            const int s = b & 1;
            if (b + 1 < NB) {
                if (b >= 1) cp.wait(t_dn[b - 1]);                   // d[1 - s] still holds batch b - 1 on its way out
                t_up[b + 1] = cp.upload(d[1 - s], h_in[b + 1], bytes);
            }
            cp.wait(t_up[b]);
            bt.data = d[s];
            ctx.check(uwip_histretch(ctx.get(), &bt, "RGB", 2, 98));
            t_dn[b] = cp.download(h_out[b], d[s], bytes, &ctx);   // starts when the stretch has finished
        }
        unsigned long long sum = 0;
        for (int b = 0; b < NB; ++b) {
            cp.wait(t_dn[b]);
            if (!cp.done(t_dn[b])) { std::printf("ticket not done after wait\n"); return 1; }
            if (std::memcmp(h_out[b], expect[b].data(), bytes) != 0) { std::printf("batch %d differs\n", b); return 1; }
            for (size_t i = 0; i < bytes; i += 97) sum += ((uint8_t *)h_out[b])[i];
        }
        for (int b = 0; b < NB; ++b) { uwip_host_free(ctx.get(), h_in[b]); uwip_host_free(ctx.get(), h_out[b]); }
        uwip_free(ctx.get(), d[0]); uwip_free(ctx.get(), d[1]);
        std::printf("copier ok %llu\n", sum);
    } catch (const uw::Error &e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
