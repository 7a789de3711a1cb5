// How much CPU a host thread (and the runtime's helper threads) burn while a thread waits for the GPU, by wait
// primitive, and what the runtime's helper thread costs per asynchronous operation of a pipeline-like loop.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/wait_cpu tools/ubench/wait_cpu.hip -lpthread
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <mutex>
#include <sys/resource.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long cycles, int *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) *out = 1;
}
static double cpu_s() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double proc_cpu_s() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + 1e-6 * r.ru_utime.tv_usec + r.ru_stime.tv_sec + 1e-6 * r.ru_stime.tv_usec; }
static double wall_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
struct latch { std::mutex m; std::condition_variable cv; bool done = false; };
static void on_done(void *p) { latch *l = (latch *)p; { std::lock_guard<std::mutex> g(l->m); l->done = true; } l->cv.notify_one(); }
static hipError_t poll_wait(hipEvent_t ev, long ns0) {
    long ns = ns0;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        timespec ts{0, ns};
        nanosleep(&ts, nullptr);
        if (ns < 1000000) ns += ns / 2;
    }
}
int main() {
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e_plain, e_block, e_notime;
    CHECK(hipEventCreate(&e_plain));
    CHECK(hipEventCreateWithFlags(&e_block, hipEventBlockingSync));
    CHECK(hipEventCreateWithFlags(&e_notime, hipEventDisableTiming));
    const long long cyc = 5000000;  // wall_clock64 ticks at 100 MHz: 50 ms
    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            const double c0 = cpu_s(), w0 = wall_s(), p0 = proc_cpu_s();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int *)nullptr);
            if (mode == 0) CHECK(hipStreamSynchronize(s));
            if (mode == 1) { CHECK(hipEventRecord(e_plain, s)); CHECK(hipEventSynchronize(e_plain)); }
            if (mode == 2) { CHECK(hipEventRecord(e_block, s)); CHECK(hipEventSynchronize(e_block)); }
            if (mode == 3) { latch l; CHECK(hipLaunchHostFunc(s, on_done, &l)); std::unique_lock<std::mutex> g(l.m); l.cv.wait(g, [&] { return l.done; }); }
            if (mode == 4) { CHECK(hipEventRecord(e_plain, s)); CHECK(poll_wait(e_plain, 50000)); }
            if (mode == 5) { CHECK(hipEventRecord(e_notime, s)); CHECK(poll_wait(e_notime, 50000)); }
            const char *names[] = {"hipStreamSynchronize", "hipEventSynchronize (default event)", "hipEventSynchronize (hipEventBlockingSync)",
                                   "hipLaunchHostFunc + condition variable", "hipEventQuery + nanosleep (default event)",
                                   "hipEventQuery + nanosleep (DisableTiming)"};
            printf("%-44s wall %6.1f ms  thread cpu %6.1f ms  process cpu %6.1f ms\n", names[mode], 1e3 * (wall_s() - w0), 1e3 * (cpu_s() - c0), 1e3 * (proc_cpu_s() - p0));
        }
    }
    // pipeline-like loops: what do the runtime's own threads cost per operation?
    const size_t bytes = 1 << 20;
    void *h_up, *h_down, *d;
    CHECK(hipHostMalloc(&h_up, bytes)); CHECK(hipHostMalloc(&h_down, bytes)); CHECK(hipMalloc(&d, bytes));
    const int iters = 400;
    const char *lnames[] = {"kernel 1 ms + event + poll", "H2D 1 MB + kernel + event + poll", "H2D + kernel + D2H 1 MB + event + poll",
                            "3 x (H2D + kernel + D2H) + event + poll", "kernel + event(DisableTiming) + poll",
                            "H2D + kernel + D2H + event(DisableTiming) + poll", "kernel + hipStreamSynchronize"};
    for (int loop = 0; loop < 7; loop++) {
        const double c0 = cpu_s(), w0 = wall_s(), p0 = proc_cpu_s();
        for (int i = 0; i < iters; i++) {
            const int reps = loop == 3 ? 3 : 1;
            for (int r = 0; r < reps; r++) {
                if (loop == 1 || loop == 2 || loop == 3 || loop == 5) CHECK(hipMemcpyAsync(d, h_up, bytes, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 100000LL, (int *)nullptr);
                if (loop == 2 || loop == 3 || loop == 5) CHECK(hipMemcpyAsync(h_down, d, bytes, hipMemcpyDeviceToHost, s));
            }
            if (loop == 6) { CHECK(hipStreamSynchronize(s)); continue; }
            hipEvent_t ev = (loop == 4 || loop == 5) ? e_notime : e_plain;
            CHECK(hipEventRecord(ev, s));
            CHECK(poll_wait(ev, 50000));
        }
        const double w = wall_s() - w0, c = cpu_s() - c0, p = proc_cpu_s() - p0;
        printf("%-52s per iteration: wall %7.1f us  caller cpu %6.1f us  other threads' cpu %6.1f us\n", lnames[loop],
               1e6 * w / iters, 1e6 * c / iters, 1e6 * (p - c) / iters);
    }
    return 0;
}
