// The ACLAHE parameter choice (modules/aclahe/python/ACLAHE.py:66-129, functions.py:49-93) ON THE DEVICE: the entropy table
// the sweep leaves in HBM never travels to the host.  One wavefront per (frame, block size) runs the knee stage of
// lm_core.hpp -- the same source as the host form uwip_aclahe_select, lanes = the 49 samples of the curve, MINPACK's
// summation order kept, so the two agree bit for bit -- and one thread per frame then makes the choice of ACLAHE.py:92-125
// (CL = the largest of the five knee indices, BS = the block size whose float16 entropy at that clip limit is largest,
// the last maximum winning).
#include "uwip_internal.hpp"
#include "lm_core.hpp"
#include <cstdlib>

namespace {

using namespace uwip_lm;

// what the kernels read besides the table: the elimination of the spline system and the derivatives of the fitted
// clip-limit axis (the axis 0.5 .. 24.5 does not depend on the image: fitted once on the host, by the same code)
struct KneeConsts {
    SplineElim E;
    double xd1[64], xd2[64];
    int x_info;
    int pad;
};

__global__ __launch_bounds__(64) void k_aclahe_knee(const float *__restrict__ tab /*[F][5][51]*/, const KneeConsts *__restrict__ kc,
                                                   int32_t *__restrict__ knee /*[F][5]*/)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    WaveLanes::vec ys, xd1, xd2;
    // the curve: clip limits 0.5 .. 24.5 = columns 1 .. 49 of the row (functions.graficar slices [2:51] of a row whose
    // column m holds cl[m - 1])
    ys.v[0] = lane < MS ? (double)tab[(size_t)b * 51 + 1 + lane] : 0.0;
    xd1.v[0] = kc->xd1[lane];
    xd2.v[0] = kc->xd2[lane];
    const int k = knee_from_curve<WaveLanes>(kc->E, ys, xd1, xd2, kc->x_info);
    if (lane == 0) knee[b] = k;
}

__device__ inline float through_half_dev(float f) { return (float)(_Float16)f; }

// par [F][4] = {BS, CL, needs an evaluation outside the swept grid, 0}
__global__ __launch_bounds__(64) void k_aclahe_choose(const float *__restrict__ tab, const int32_t *__restrict__ knee, int F,
                                                     int32_t *__restrict__ par, int force_d)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    int d = -1;
    for (int g = 0; g < 5; ++g) d = max(d, knee[(size_t)f * 5 + g]);
    if (d < 0) d = 0;
    if (force_d >= 0) d = force_d;                  // test hook (UWIP_ACLAHE_TEST_FORCE_CL): exercises the out-of-grid path
    const bool outside = 2 * d > 50;
    int w = 0;
    float best = 0.f;
    for (int g = 0; g < 5; ++g) {
        const float h = through_half_dev(tab[((size_t)f * 5 + g) * 51 + (outside ? 50 : 2 * d)]);
        if (g == 0 || h >= best) { best = h; w = g; }      // last maximum wins (ACLAHE.py:118-124)
    }
    const int BlockSize[5] = {2, 4, 8, 16, 32};
    par[(size_t)f * 4 + 0] = BlockSize[w];
    par[(size_t)f * 4 + 1] = d;
    par[(size_t)f * 4 + 2] = outside ? 1 : 0;
    par[(size_t)f * 4 + 3] = 0;
}

}  // namespace

// aclahe_select.cpp: the host form's constants (the same code fitted the axis)
void uwip_aclahe_knee_consts(uwip_lm::SplineElim *E, double *xd1, double *xd2, int *x_info);

// d_entropy [frames][5][51] (device) -> d_par [frames][4] int32 {BS, CL, need_eval, 0} and, optionally, d_knee [frames][5]
UWIP_API int uwip_aclahe_select_device(uwip_ctx *ctx, const float *d_entropy, int frames, int32_t *d_par, int32_t *d_knee)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (frames <= 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, d_entropy && d_par, "null buffer");
    size_t bytes = 0;
    const void *d_kc = uwip_table_find(ctx, "aclahe.knee_consts", &bytes);
    if (!d_kc) {
        KneeConsts h{};
        uwip_aclahe_knee_consts(&h.E, h.xd1, h.xd2, &h.x_info);
        d_kc = uwip_table_put(ctx, "aclahe.knee_consts", &h, sizeof h);
        if (!d_kc) return UWIP_ERR_NOMEM;
    }
    int32_t *knee = d_knee ? d_knee : (int32_t *)uwip_ws(ctx, "auto.knee", sizeof(int32_t) * 5 * (size_t)frames);
    if (!knee) return UWIP_ERR_NOMEM;
    {
        uwip_kscope ks(ctx, "k_aclahe_knee");
        k_aclahe_knee<<<frames * 5, 64, 0, ctx->stream>>>(d_entropy, (const KneeConsts *)d_kc, knee);
        UWIP_HIP(ctx, hipGetLastError());
    }
    {
        uwip_kscope ks(ctx, "k_aclahe_choose");
        // a knee index >= 26 takes a degenerate fit: the tests force one to reach the exact block-size search
        // (a test hook: only in a process started with UWIP_TEST_HOOKS=1, uwip_internal.hpp)
        const char *fe = uwip_test_hooks() ? std::getenv("UWIP_ACLAHE_TEST_FORCE_CL") : nullptr;
        const int force_d = fe && *fe ? std::atoi(fe) : -1;
        k_aclahe_choose<<<uwip_cdiv(frames, 64), 64, 0, ctx->stream>>>(d_entropy, knee, frames, d_par, force_d);
        UWIP_HIP(ctx, hipGetLastError());
    }
    return UWIP_OK;
}
