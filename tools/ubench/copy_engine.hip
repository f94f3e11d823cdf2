// Which engine moves a big pinned-host <-> HBM copy (SDMA, seen by rocprofv3 --memory-copy-trace, or the runtime's blit
// kernel __amd_rocclr_copyBuffer, seen by --kernel-trace) and at what rate, in the situations the host-buffer front end
// creates.  Every variant is one 398 MB copy, separated by a device synchronise; the n-th big record of the trace is the
// n-th variant printed here.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/copy_engine.hip -o tools/ubench/copy_engine.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(float *p, int iters)
{
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t n = (size_t)64 * 1080 * 1920 * 3;
    void *d0, *d1, *h0, *h1, *hr;
    float *scratch;
    CK(hipMalloc(&d0, n)); CK(hipMalloc(&d1, n)); CK(hipMalloc(&scratch, 1 << 20));
    CK(hipHostMalloc(&h0, n, hipHostMallocDefault));
    CK(hipHostMalloc(&h1, n, hipHostMallocNonCoherent));
    hr = aligned_alloc(4096, n); memset(hr, 1, n);
    CK(hipHostRegister(hr, n, hipHostRegisterDefault));
    memset(h0, 1, n); memset(h1, 1, n);
    hipStream_t A, B, C;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int iters = 4000000;       // a few ms of one tiny block
    auto run = [&](const char *name, auto fn) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        fn();
        CK(hipDeviceSynchronize());
        const double dt = now() - t0;
        printf("%-58s %7.2f ms  %6.1f GB/s\n", name, dt * 1e3, n / dt / 1e9);
        fflush(stdout);
    };
    run("warm D2H", [&] { CK(hipMemcpyAsync(h0, d0, n, hipMemcpyDeviceToHost, B)); });
    run("warm H2D", [&] { CK(hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, B)); });
    run("1 D2H coherent pinned, idle stream", [&] { CK(hipMemcpyAsync(h0, d0, n, hipMemcpyDeviceToHost, B)); });
    run("2 H2D coherent pinned, idle stream", [&] { CK(hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, B)); });
    run("3 D2H non-coherent pinned, idle stream", [&] { CK(hipMemcpyAsync(h1, d0, n, hipMemcpyDeviceToHost, B)); });
    run("4 D2H hipHostRegister'ed, idle stream", [&] { CK(hipMemcpyAsync(hr, d0, n, hipMemcpyDeviceToHost, B)); });
    run("5 D2H after StreamWaitEvent on another stream's kernel", [&] {
        spin<<<1, 64, 0, A>>>(scratch, iters / 8); CK(hipEventRecord(ev, A)); CK(hipStreamWaitEvent(B, ev, 0));
        CK(hipMemcpyAsync(h0, d0, n, hipMemcpyDeviceToHost, B)); });
    run("6 H2D after StreamWaitEvent on another stream's kernel", [&] {
        spin<<<1, 64, 0, A>>>(scratch, iters / 8); CK(hipEventRecord(ev, A)); CK(hipStreamWaitEvent(B, ev, 0));
        CK(hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, B)); });
    run("7 D2H behind a kernel of the same stream", [&] {
        spin<<<1, 64, 0, B>>>(scratch, iters / 8); CK(hipMemcpyAsync(h0, d0, n, hipMemcpyDeviceToHost, B)); });
    run("8 H2D behind a kernel of the same stream", [&] {
        spin<<<1, 64, 0, B>>>(scratch, iters / 8); CK(hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, B)); });
    run("9 D2H + H2D at once on two streams (two records)", [&] {
        CK(hipMemcpyAsync(h0, d0, n, hipMemcpyDeviceToHost, B)); CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, C)); });
    run("10 two D2H at once on two streams (two records)", [&] {
        CK(hipMemcpyAsync(h0, d0, n, hipMemcpyDeviceToHost, B)); CK(hipMemcpyAsync(h1, d1, n, hipMemcpyDeviceToHost, C)); });
    run("11 two H2D at once on two streams (two records)", [&] {
        CK(hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, B)); CK(hipMemcpyAsync(d1, h1, n, hipMemcpyHostToDevice, C)); });
    run("12 D2H in 16 chunks of 25 MB on one stream (16 records)", [&] {
        for (int i = 0; i < 16; ++i) CK(hipMemcpyAsync((char *)h0 + i * (n / 16), (char *)d0 + i * (n / 16), n / 16, hipMemcpyDeviceToHost, B)); });
    run("13 D2H via hipMemcpyDtoHAsync", [&] { CK(hipMemcpyDtoHAsync(h0, (hipDeviceptr_t)d0, n, B)); });
    return 0;
}
