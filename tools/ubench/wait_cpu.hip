// How much host CPU does waiting for the GPU cost?  A ~100 ms kernel, waited for in four ways; prints the waiting
// thread's CPU time (CLOCK_THREAD_CPUTIME_ID) beside the wall time.
//   hipcc --offload-arch=gfx950 -O2 -o wait_cpu.bin wait_cpu.hip && ./wait_cpu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <thread>
#include <vector>

__global__ void spin_kernel(long long cycles, int *out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}

static double now(clockid_t c)
{
    timespec ts;
    clock_gettime(c, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

int main(int argc, char **argv)
{
    const bool sched_blocking = argc > 1 && argv[1][0] == 'b';
    if (sched_blocking) CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t evb, evd;
    CK(hipEventCreateWithFlags(&evb, hipEventDisableTiming | hipEventBlockingSync));
    CK(hipEventCreateWithFlags(&evd, hipEventDisableTiming));
    const long long cyc = 10000000;          // wall_clock64 ticks at 100 MHz: 100 ms
    spin_kernel<<<1, 64, 0, st>>>(1000, nullptr);
    CK(hipStreamSynchronize(st));
    std::printf("device flags: %s\n", sched_blocking ? "hipDeviceScheduleBlockingSync" : "default");
    for (int mode = 0; mode < 4; ++mode) {
        spin_kernel<<<1, 64, 0, st>>>(cyc, nullptr);
        const double w0 = now(CLOCK_MONOTONIC), c0 = now(CLOCK_THREAD_CPUTIME_ID), p0 = now(CLOCK_PROCESS_CPUTIME_ID);
        const char *name = "";
        if (mode == 0) { name = "hipStreamSynchronize"; CK(hipStreamSynchronize(st)); }
        if (mode == 1) { name = "hipEventSynchronize (hipEventBlockingSync)"; CK(hipEventRecord(evb, st)); CK(hipEventSynchronize(evb)); }
        if (mode == 2) { name = "hipEventSynchronize (default event)"; CK(hipEventRecord(evd, st)); CK(hipEventSynchronize(evd)); }
        if (mode == 3) {
            name = "hipEventQuery + 200 us sleeps";
            CK(hipEventRecord(evd, st));
            while (hipEventQuery(evd) == hipErrorNotReady) { timespec ts{0, 200000}; nanosleep(&ts, nullptr); }
        }
        const double w = now(CLOCK_MONOTONIC) - w0, c = now(CLOCK_THREAD_CPUTIME_ID) - c0, p = now(CLOCK_PROCESS_CPUTIME_ID) - p0;
        std::printf("  %-46s wall %7.2f ms  thread CPU %7.2f ms  process CPU %7.2f ms\n", name, w * 1e3, c * 1e3, p * 1e3);
    }
    return 0;
}
