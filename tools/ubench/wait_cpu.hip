// How much CPU a host thread burns while it waits for a ~50 ms kernel, by wait primitive.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/wait_cpu tools/ubench/wait_cpu.hip -lpthread
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdio>
#include <ctime>
#include <mutex>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long cycles, int *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out) *out = 1;
}
static double cpu_s() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#include <sys/resource.h>
static double proc_cpu_s() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + 1e-6 * r.ru_utime.tv_usec + r.ru_stime.tv_sec + 1e-6 * r.ru_stime.tv_usec; }
static double wall_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
struct latch { std::mutex m; std::condition_variable cv; bool done = false; };
static void on_done(void *p) { latch *l = (latch *)p; { std::lock_guard<std::mutex> g(l->m); l->done = true; } l->cv.notify_one(); }
int main() {
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e_plain, e_block;
    CHECK(hipEventCreate(&e_plain));
    CHECK(hipEventCreateWithFlags(&e_block, hipEventBlockingSync));
    const long long cyc = 5000000;  // wall_clock64 ticks at 100 MHz: 50 ms
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            const double c0 = cpu_s(), w0 = wall_s(), p0 = proc_cpu_s();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int *)nullptr);
            if (mode == 0) CHECK(hipStreamSynchronize(s));
            if (mode == 1) { CHECK(hipEventRecord(e_plain, s)); CHECK(hipEventSynchronize(e_plain)); }
            if (mode == 2) { CHECK(hipEventRecord(e_block, s)); CHECK(hipEventSynchronize(e_block)); }
            if (mode == 3) { latch l; CHECK(hipLaunchHostFunc(s, on_done, &l)); std::unique_lock<std::mutex> g(l.m); l.cv.wait(g, [&] { return l.done; }); }
            const char *names[] = {"hipStreamSynchronize", "hipEventSynchronize (default event)", "hipEventSynchronize (hipEventBlockingSync)", "hipLaunchHostFunc + condition variable"};
            printf("%-44s wall %6.1f ms  thread cpu %6.1f ms  process cpu %6.1f ms\n", names[mode], 1e3 * (wall_s() - w0), 1e3 * (cpu_s() - c0), 1e3 * (proc_cpu_s() - p0));
        }
    }
    return 0;
}
